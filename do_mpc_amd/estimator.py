"""Moving horizon estimation with do_mpc.estimator.MHE's user surface on the MI355X IPM backend.

Reference surface mirrored here (same names, argument meaning and error behaviour):
  MHE(model, p_est_list)                       /root/reference/do_mpc/estimator/_mhe.py:150-233
  set_objective / set_default_objective        :489-716
  get_p_template / set_p_fun / get_y_template / set_y_fun / get_tvp_template / set_tvp_fun   :717-801, optimizer.py:588-676
  bounds / scaling / set_nl_cons               optimizer.py:268-541 (with `_p_est` as a variable type)
  setup / set_initial_guess / make_step        :864-993
  opt_x = [_x | _z | _u | _w | _v | _eps | _p_est], opt_p = [_x_prev | _p_est_prev | _p_set | _tvp | _y_meas]   :1052-1094

Below the surface the estimation problem runs on the kernels of the controller (structured interior point: per-interval
elimination + Riccati recursion over the horizon, csrc/).  Its NLP (_mhe.py:1030-1211) is a chain like the controller's with
four differences, each a switch of the generated header (csrc/dompc_kernel.h: DOMPC_FREE_ROOT / DOMPC_LT_END / DOMPC_NL_DUP):
  * the initial state is FREE and carries the arrival cost; the previous estimate sits where the controller has x0;
  * the estimated parameters `_p_est` are ONE variable for the whole horizon.  Here they ride as additional states with
    d/dt = 0: every stored point of the horizon has its own copy, tied together by the (linear) collocation and continuity
    rows of those states - the same feasible set, objective and barrier terms in the reduced space, hence the same solution
    (the interior-point iterates differ from IPOPT's on the reference's formulation; solutions agree to the solver tolerance);
  * the measurement rows  h(x_{k+1}, u_k, tvp_k, p) + v_k = y_k  are solved for the measurement noise v_k, which only
    appears in the stage cost: the cost of stage k becomes a function of the END state of the interval.  A measurement
    WITHOUT noise (`set_meas(..., meas_noise=False)`) is supported where it measures an input variable as it is - the use the
    reference's documentation suggests (_model.py:693-695): the rows  u - y = 0  fix that input, which leaves the chain problem
    (it becomes an entry of the stage's time-varying parameters); `opt_x['_u']` hands it back and the multipliers of those rows
    are the derivatives of the chain problem's Lagrangian w.r.t. it (other noise-free measurements are refused by name);
  * process noise `_w` is one more input of the interval;
  * algebraic states `_z` of the model (and its algebraic equations) are edge unknowns of the chain problem exactly as in the
    controller; the measurement function reads `_z[k, -1]` (:1158) - what a stage cost reads there.  Discrete-time models: the
    next state itself rides as an algebraic state of the interval (no stored points inside an interval otherwise).
`opt_x_num`, `opt_p_num`, `lam_g_num` are handed out in the reference's layout.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import numpy as np

from . import sym
from .controller import MPC, MPCData, _Indexed
from .model import Model, VarGroup
from .structs import Entry, Layout, NumStruct


@dataclass
class MHESettings:
    n_horizon: int = None
    t_step: float = None
    meas_from_data: bool = True
    state_discretization: str = "collocation"
    collocation_type: str = "radau"
    collocation_deg: int = 2
    collocation_ni: int = 1
    nl_cons_check_colloc_points: bool = False
    nl_cons_single_slack: bool = False
    cons_check_colloc_points: bool = True
    store_full_solution: bool = False
    store_lagr_multiplier: bool = True
    store_solver_stats: List[str] = field(default_factory=lambda: ["success", "t_wall_total"])
    nlpsol_opts: Dict = field(default_factory=dict)
    gpu_index: int = 0
    max_batch: int = 1
    block_threads: int = 0

    def check_for_mandatory_settings(self):
        if self.n_horizon is None:
            raise ValueError("n_horizon must be set")
        if self.t_step is None:
            raise ValueError("t_step must be set")

    def supress_ipopt_output(self):
        self.nlpsol_opts.update({"ipopt.print_level": 0, "ipopt.sb": "yes", "print_time": 0})

    def set_linear_solver(self, solver_name: str = "MA27"):
        self.nlpsol_opts.update({"ipopt.linear_solver": solver_name})


def _arr(v) -> np.ndarray:
    return np.asarray(v.master if hasattr(v, "master") else v, dtype=float).reshape(-1)


def _group(kind: str, items) -> VarGroup:
    g = VarGroup(kind)
    for n, v in items:
        g.add(n, v)
    return g


class MHE:
    def __init__(self, model: Model, p_est_list: list = [], settings: Optional[MHESettings] = None):
        assert model.flags["setup"] is True, "Model for MHE was not setup. After the complete model creation call model.setup()."
        self.model = m = model
        self.settings = self._settings = settings if settings is not None else MHESettings()
        pn = [n for n in m._p.names if m._p.vars[n].numel() > 0]
        for n in p_est_list:
            assert n in pn, "The item {} in p_est_list is not a parameter of the model".format(n)
        self._p_est = _group("_p_est", [(n, m._p.vars[n]) for n in pn if n in p_est_list])
        self._p_set = _group("_p_set", [(n, m._p.vars[n]) for n in pn if n not in p_est_list])
        self._p_est_prev = _group("_p_est_prev", [(n, sym.SX.sym(n + "_prev", *m._p.vars[n].shape)) for n in self._p_est.names])
        self._x_prev = _group("_x_prev", [(n, sym.SX.sym(n + "_prev", *m._x.vars[n].shape)) for n in m._x.names])
        self._x, self._w, self._v = m._x, m._w, m._v
        self.n_p_est, self.n_p_set = self._p_est.size, self._p_set.size
        self._x_lb, self._x_ub = m._x(-np.inf), m._x(np.inf)
        self._u_lb, self._u_ub = m._u(-np.inf), m._u(np.inf)
        self._z_lb, self._z_ub = m._z(-np.inf), m._z(np.inf)
        self._p_est_lb, self._p_est_ub = self._p_est(-np.inf), self._p_est(np.inf)
        self._x_scaling, self._u_scaling, self._z_scaling = m._x(1.0), m._u(1.0), m._z(1.0)
        self._p_est_scaling, self._p_set_scaling = self._p_est(1.0), self._p_set(1.0)
        self._x0, self._u0, self._z0, self._t0 = m._x(0.0), m._u(0.0), m._z(0.0), np.array([0.0])
        self._p_est0 = self._p_est(0.0)
        self.nl_cons_list: List[dict] = []
        self.flags = {"setup": False, "set_objective": False, "set_tvp_fun": False, "set_p_fun": False, "set_y_fun": False,
                      "set_initial_guess": False}
        self.bounds = _Indexed(self._get_bounds, self._set_bounds)
        self.scaling = _Indexed(self._get_scaling, self._set_scaling)
        self.data = MPCData(model)
        self.solver_stats: dict = {}

    # ------------------------------------------------------------------ settings, bounds, scaling
    def set_param(self, **kwargs) -> None:
        assert self.flags["setup"] is False, "Setting parameters after setup is prohibited."
        for key, value in kwargs.items():
            if hasattr(self.settings, key):
                setattr(self.settings, key, value)
            else:
                print("Warning: Key {} does not exist for MHE.".format(key))

    def _bound_struct(self, ind):
        assert isinstance(ind, tuple) and len(ind) >= 2, "Power index must include bound_type, var_type, var_name (as a tuple)."
        bound_type, var_type, var_name = ind[0], ind[1], ind[2:]
        if bound_type not in ("lower", "upper"):
            raise Exception("Invalid power index {} for bound_type. Must be from (lower, upper).".format(bound_type))
        tab = {("lower", "_x"): self._x_lb, ("upper", "_x"): self._x_ub, ("lower", "_u"): self._u_lb, ("upper", "_u"): self._u_ub,
               ("lower", "_z"): self._z_lb, ("upper", "_z"): self._z_ub, ("lower", "_p_est"): self._p_est_lb, ("upper", "_p_est"): self._p_est_ub}
        if (bound_type, var_type) not in tab:
            raise Exception("Invalid power index {} for var_type. Must be from (_x, _u, _z, _p_est).".format(var_type))
        return tab[(bound_type, var_type)], var_name

    def _get_bounds(self, ind):
        st, name = self._bound_struct(ind)
        return st[name] if name else st

    def _set_bounds(self, ind, val):
        st, name = self._bound_struct(ind)
        st[name] = val

    def _scaling_struct(self, ind):
        tab = {"_x": self._x_scaling, "_u": self._u_scaling, "_z": self._z_scaling, "_p_est": self._p_est_scaling, "_p_set": self._p_set_scaling}
        if ind[0] not in tab:
            raise Exception("Invalid power index {} for var_type. Must be from (_x, _u, _z, _p_est, _p_set).".format(ind[0]))
        return tab[ind[0]], ind[1:]

    def _get_scaling(self, ind):
        st, name = self._scaling_struct(ind)
        return st[name] if name else st

    def _set_scaling(self, ind, val):
        st, name = self._scaling_struct(ind)
        st[name] = val

    def _set_iter(self, attr, val):
        st = getattr(self, attr)
        if hasattr(val, "master"):
            val = val.master
        val = np.asarray(val, dtype=float).reshape(-1)
        if val.size == 1 and st.size > 1:
            val = np.full(st.size, float(val[0]))
        assert val.size == st.size, "Wrong input with shape {}. Expected vector with {} elements".format(val.shape, st.size)
        st.master[:] = val

    x0 = property(lambda self: self._x0, lambda self, v: self._set_iter("_x0", v))
    u0 = property(lambda self: self._u0, lambda self, v: self._set_iter("_u0", v))
    z0 = property(lambda self: self._z0, lambda self, v: self._set_iter("_z0", v))
    p_est0 = property(lambda self: self._p_est0, lambda self, v: self._set_iter("_p_est0", v))
    t0 = property(lambda self: self._t0, lambda self, v: setattr(self, "_t0", np.array(v, dtype=float).reshape(-1)[:1]))

    # ------------------------------------------------------------------ objective (_mhe.py:489-716)
    def set_objective(self, stage_cost, arrival_cost) -> None:
        stage_cost, arrival_cost = sym.SX(stage_cost), sym.SX(arrival_cost)
        assert stage_cost.shape == (1, 1), "stage_cost must have shape=(1,1). You have {}".format(stage_cost.shape)
        assert arrival_cost.shape == (1, 1), "arrival_cost must have shape=(1,1). You have {}".format(arrival_cost.shape)
        assert self.flags["setup"] is False, "Cannot call .set_objective after .setup."
        m = self.model
        ok_stage = self._w.cat.nodes() + self._v.cat.nodes() + m._tvp.cat.nodes() + m._p.cat.nodes()
        if [s for s in sym.free_symbols(stage_cost.nodes()) if s.idx not in {q.idx for q in ok_stage}]:
            raise Exception("objective cost equation must be solely depending on w, v, p and tvp.")
        ok_arr = m._x.cat.nodes() + self._x_prev.cat.nodes() + self._p_est_prev.cat.nodes() + m._p.cat.nodes()
        if [s for s in sym.free_symbols(arrival_cost.nodes()) if s.idx not in {q.idx for q in ok_arr}]:
            raise Exception("Arrival cost equation must be solely depending on x_0, x_prev, p_0, p_prev, p_set")
        self.stage_cost, self.arrival_cost = stage_cost, arrival_cost
        self.flags["set_objective"] = True

    def set_default_objective(self, P_x, P_v=None, P_p=None, P_w=None) -> None:
        m = self.model
        n_x, n_w, n_v, n_p = m.n_x, m.n_w, m.n_v, self.n_p_est
        as_sx = lambda P: P if isinstance(P, sym.SX) else sym.SX(np.asarray(P, dtype=float))      # noqa: E731
        assert np.shape(P_x) == (n_x, n_x) or getattr(P_x, "shape", None) == (n_x, n_x), "P_x has wrong shape, must be {}".format((n_x, n_x))
        stage_cost = sym.SX(0.0)
        if P_v is None:
            assert n_v == 0, "Must pass weighting factor P_v, since you have measurement noise on some measurements (configured in model)."
        else:
            v = self._v.cat
            stage_cost = stage_cost + v.T @ as_sx(P_v) @ v
        if P_w is None:
            assert n_w == 0, "Must pass weighting factor P_w, since you have process noise on some states (configured in model)."
        else:
            w = self._w.cat
            stage_cost = stage_cost + w.T @ as_sx(P_w) @ w
        dx = self._x.cat - self._x_prev.cat
        arrival_cost = dx.T @ as_sx(P_x) @ dx
        if P_p is None:
            assert n_p == 0, "Must pass weighting factor P_p, since you are trying to estimate parameters."
        else:
            dp = self._p_est.cat - self._p_est_prev.cat
            Pp = as_sx(P_p)
            arrival_cost = arrival_cost + dp.T @ (Pp.reshape((n_p, n_p)) if Pp.numel() == n_p * n_p else Pp) @ dp
        self.set_objective(stage_cost, arrival_cost)

    def set_nl_cons(self, expr_name: str, expr, ub: float = np.inf, soft_constraint: bool = False, penalty_term_cons: float = 1.0,
                    maximum_violation: float = np.inf):
        assert self.flags["setup"] is False, "Cannot call .set_expression after .setup_model."
        self.nl_cons_list.append(dict(expr_name=expr_name, expr=expr, ub=ub, soft_constraint=soft_constraint,
                                      penalty_term_cons=penalty_term_cons, maximum_violation=maximum_violation))
        return sym.SX(expr)

    # ------------------------------------------------------------------ templates / time-varying data (_mhe.py:717-801)
    def get_p_template(self) -> NumStruct:
        return self._p_set(0.0)

    def set_p_fun(self, p_fun: Callable) -> None:
        assert self.get_p_template().labels() == p_fun(0).labels(), "Incorrect output of p_fun. Use get_p_template to obtain the required structure."
        self.p_fun = p_fun
        self.flags["set_p_fun"] = True

    def get_tvp_template(self) -> NumStruct:
        return NumStruct(Layout([Entry("_tvp", struct=self.model._tvp.layout(), repeat=self.settings.n_horizon)]), 0.0)

    def set_tvp_fun(self, tvp_fun: Callable) -> None:
        assert self.get_tvp_template().labels() == tvp_fun(0).labels(), "Incorrect output of tvp_fun. Use get_tvp_template to obtain the required structure."
        self.tvp_fun = tvp_fun
        self.flags["set_tvp_fun"] = True

    def get_y_template(self) -> NumStruct:
        return NumStruct(Layout([Entry("y_meas", struct=self.model._y.layout(), repeat=self.settings.n_horizon)]), 0.0)

    def set_y_fun(self, y_fun: Callable) -> None:
        assert self.get_y_template().labels() == y_fun(0).labels(), "Incorrect output of y_fun. Use get_y_template to obtain the required structure."
        self.y_fun = y_fun
        self.flags["set_y_fun"] = True

    def _check_validity(self):
        if not self.flags["set_objective"]:
            raise Exception("Objective is undefined. Please call .set_objective() or .set_default_objective() prior to .setup().")
        if not self.flags["set_tvp_fun"]:
            if self.model.n_tvp:
                raise Exception("You have not supplied a function to obtain the time-varying parameters defined in model. Use .set_tvp_fun() prior to setup.")
            _tvp = self.get_tvp_template()
            self.set_tvp_fun(lambda t: _tvp)
        if not self.flags["set_p_fun"]:
            if self.n_p_set:
                raise Exception("You have not supplied a function to obtain the parameters defined in model. Use .set_p_fun() prior to setup.")
            _p = self.get_p_template()
            self.set_p_fun(lambda t: _p)
        if not self.flags["set_y_fun"]:
            if not self.settings.meas_from_data:
                raise Exception("You have not suppplied a measurement function. Use .set_y_fun or set parameter meas_from_data to True for default function.")
            y_template = self.get_y_template()

            def y_fun(t_now):           # (the last measurements of the data object, the oldest entries left at zero: _mhe.py:845-860)
                n_steps = min(self.data["_y"].shape[0], self.settings.n_horizon)
                for k in range(-n_steps, 0):
                    y_template["y_meas", k] = self.data["_y"][k]
                return y_template
            self.set_y_fun(y_fun)

    # ------------------------------------------------------------------ setup
    def setup(self) -> None:
        s, m = self.settings, self.model
        s.check_for_mandatory_settings()
        self._check_validity()
        nx, nu, nw, nv, ny, npe = m.n_x, m.n_u, m.n_w, m.n_v, m.n_y, self.n_p_est
        N = s.n_horizon
        # (scaling of the estimated parameters: they ride as states of the augmented model, below; a scaling of the FIXED parameters
        #  cancels in the reference's NLP - `_p_set / _p_set_scaling` times the model's `_p_scaling`, _mhe.py:1127, 1040 - and is ignored)
        # ---- measurement noise as a function of (x, u, tvp, p, y_meas).  A measurement WITH its noise term (the default) is solved for
        # it: v_k = y_k - h(x_{k+1}, u_k, ...).  A measurement WITHOUT noise (`set_meas(..., meas_noise=False)`, _model.py:670-735) is an
        # equality row  h - y = 0  (_mhe.py:1144-1158).  Supported where the reference's documentation suggests it ("deactivate
        # measurement noise for measured inputs: certain variables", _model.py:693-695; the MHE example notebook): the measurement of an
        # INPUT variable as it is.  That input then is no variable of the chain problem - it is replaced by its measurement (one entry of
        # the time-varying parameters) everywhere; opt_x['_u'] hands it back, the multiplier of its row is the derivative of the chain
        # problem's Lagrangian w.r.t. that parameter (_fixed_input_multipliers).
        y_sym = sym.SX.sym("y_meas", ny, 1)
        zero_v = sym.SX(np.zeros((nv, 1)))
        h0 = sym.substitute(m._y.cat, m._v.cat, zero_v) if nv else m._y.cat
        dy_dv = sym.jacobian(m._y.cat, m._v.cat) if nv else None
        D = dy_dv.to_numpy().reshape(ny, nv) if (nv and dy_dv.is_constant()) else None
        if nv and D is None:
            raise NotImplementedError("structured HIP backend: the measurement noise must enter the measurement equations additively")
        noisy = [i for i in range(ny) if nv and np.any(D[i] != 0.0)]
        if len(noisy) != nv or (nv and not np.array_equal(D[noisy], np.eye(nv))):
            raise NotImplementedError("structured HIP backend: every noisy measurement of the estimator model needs its own additive "
                                      "noise term: the measurement rows are solved for it")
        self._y_noisy = np.array(noisy, dtype=int)
        self._y_free = np.array([i for i in range(ny) if i not in noisy], dtype=int)
        u_nodes = m._u.cat.nodes()
        u_of_row = {}
        for i in self._y_free:
            j = next((j for j, un in enumerate(u_nodes) if un.idx == h0.nodes()[i].idx), None)
            if j is None or j in u_of_row.values():
                raise NotImplementedError("structured HIP backend: a measurement without noise (set_meas(..., meas_noise=False)) is "
                                          "supported where it measures an input variable as it is (each input once)")
            u_of_row[int(i)] = j
        fixed_el = set(u_of_row.values())
        off, self._u_fixed_names = 0, []
        for n in m._u.names:
            k = m._u.vars[n].numel()
            hit = [j in fixed_el for j in range(off, off + k)]
            if k and any(hit):
                if not all(hit):
                    raise NotImplementedError("structured HIP backend: input '{}' is measured without noise in some of its elements "
                                              "only".format(n))
                self._u_fixed_names.append(n)
            off += k
        self._u_keep = np.array([j for j in range(nu) if j not in fixed_el], dtype=int)
        self._u_fixed = np.array([u_of_row[int(i)] for i in self._y_free], dtype=int)       # input element of the free measurement rows, in row order
        nuk = self._nuk = self._u_keep.size
        if self._u_fixed.size:
            if nv == 0:
                raise NotImplementedError("structured HIP backend: an estimator whose measurements are ALL without noise")
            if m.model_type == "discrete" or m.n_z:
                raise NotImplementedError("structured HIP backend: measurements without noise for discrete-time models / models with "
                                          "algebraic states")
            u_repl = sym.vertcat(*[(y_sym[int(self._y_free[list(self._u_fixed).index(j)])] if j in fixed_el else m._u.cat[j]) for j in range(nu)])
            fix = lambda e: sym.substitute(sym.SX(e), m._u.cat, u_repl)      # noqa: E731
            h0 = fix(h0)
            if any(sym.depends_on(sym.SX(c["expr"]).nodes(), [u_nodes[j] for j in fixed_el]) for c in self.nl_cons_list):
                raise NotImplementedError("structured HIP backend: nl_cons rows that depend on an input measured without noise")
        else:
            fix = lambda e: e                                                  # noqa: E731
        discrete = self._discrete = m.model_type == "discrete"
        nz = m.n_z
        if discrete:
            # a discrete model has no stored points inside an interval: the NEXT state becomes an algebraic state of the interval,
            # z = f(x, u, w, p) as its algebraic equation and x+ = z as its dynamics - the measurement residual (and with it the stage
            # cost) then reads that algebraic state, which the dense edge path of the DAE models handles (`_z[k, s, -1]` in the cost)
            z_next = sym.SX.sym("x_next", nx, 1)
            h0 = sym.substitute(h0, m._x.cat, z_next)
        v_of = (y_sym - h0)[[int(i) for i in self._y_noisy]] if self._y_free.size else y_sym - h0       # v_k = y_k - h(x_{k+1}, u_k, tvp_k, p): the noisy rows
        stage = sym.substitute(self.stage_cost, m._v.cat, v_of) if nv else self.stage_cost
        # ---- the augmented model: states (x, p_est), inputs (u, w), tvp (tvp, y_meas), parameters p_set; symbols are shared
        am = Model("discrete" if discrete else "continuous")
        for n in m._x.names:
            am._x.add(n, m._x.vars[n])
        for n in self._p_est.names:
            am._x.add(n, self._p_est.vars[n])
        for n in m._u.names:
            if m._u.vars[n].numel() and n not in self._u_fixed_names:
                am._u.add(n, m._u.vars[n])
        for n in m._w.names:
            if m._w.vars[n].numel():
                am._u.add(n, m._w.vars[n])
        for n in m._tvp.names:
            if m._tvp.vars[n].numel():
                am._tvp.add(n, m._tvp.vars[n])
        am._tvp.add("y_meas", y_sym)
        for n in self._p_set.names:
            am._p.add(n, self._p_set.vars[n])
        if discrete:
            # (the model's own algebraic states and equations first - the reference's rows of a discrete interval are [alg ; rhs - x+],
            #  optimizer.py:820-824 - then the copy of the next state)
            for n in m._z.names:
                if m._z.vars[n].numel():
                    am._z.add(n, m._z.vars[n])
            for a in m.alg_list:
                am.alg_list.append(dict(a))
            am._z.add("x_next", z_next)
            am.alg_list.append({"expr_name": "x_next", "expr": z_next - m._rhs})
            off = 0
            for n in m._x.names:
                k = m._x.vars[n].numel()
                am.rhs_list.append({"var_name": n, "expr": z_next[off:off + k].reshape(m._x.vars[n].shape)})
                off += k
            for n in self._p_est.names:
                am.rhs_list.append({"var_name": n, "expr": self._p_est.vars[n]})          # p+ = p
        else:
            for r in m.rhs_list:
                am.rhs_list.append(dict(r, expr=fix(r["expr"])))
            for n in self._p_est.names:
                am.rhs_list.append({"var_name": n, "expr": sym.SX(np.zeros(self._p_est.vars[n].shape))})
            # algebraic states of the model (_mhe.py:1056, 1136-1141): edge unknowns of the chain problem like in the controller; the
            # measurement function reads `_z[k, -1]` (:1158), which is what a stage cost reads there (dense edge path)
            for n in m._z.names:
                if m._z.vars[n].numel():
                    am._z.add(n, m._z.vars[n])
            for a in m.alg_list:
                am.alg_list.append(dict(a))
        for n in m._aux.names:
            if n != "default":
                am._aux.add(n, fix(m._aux.vars[n]))
        am.setup()
        self._aug_model = am
        # ---- the chain problem on the controller's machinery
        mpc = MPC(am)
        st = mpc.settings
        st.n_horizon, st.t_step, st.n_robust, st.open_loop = N, s.t_step, 0, False
        st.state_discretization, st.collocation_type = s.state_discretization, s.collocation_type
        st.collocation_deg, st.collocation_ni = s.collocation_deg, s.collocation_ni
        st.nl_cons_check_colloc_points, st.cons_check_colloc_points = s.nl_cons_check_colloc_points, s.cons_check_colloc_points
        st.nl_cons_single_slack = bool(s.nl_cons_single_slack)      # (_mhe.py:1046-1049, 1161: one `_eps` entry for all stages - shared variables, csrc: EPS_GLOBAL)
        st.store_full_solution, st.nlpsol_opts = False, dict(s.nlpsol_opts)
        st.gpu_index, st.max_batch, st.block_threads = s.gpu_index, s.max_batch, s.block_threads
        mpc.set_objective(mterm=sym.SX(0.0), lterm=stage)
        mpc.set_rterm(**{n: 0.0 for n in am._u.names if am._u.vars[n].numel()})
        for c in self.nl_cons_list:
            mpc.set_nl_cons(c["expr_name"], c["expr"], ub=c["ub"], soft_constraint=c["soft_constraint"],
                            penalty_term_cons=c["penalty_term_cons"], maximum_violation=c["maximum_violation"])
        mpc._x_scaling.master[:nx] = self._x_scaling.master
        mpc._x_scaling.master[nx:nx + npe] = self._p_est_scaling.master          # (_mhe.py:1083)
        self._sx_aug = mpc._x_scaling.master.copy()                               # scaling of the augmented state (x, p_est)
        mpc._u_scaling.master[:nuk] = self._u_scaling.master[self._u_keep]
        if nz:
            mpc._z_scaling.master[:nz] = self._z_scaling.master
        mpc.set_tvp_fun(lambda t: mpc.get_tvp_template())
        if am.n_p:
            mpc.set_p_fun(lambda t: mpc.get_p_template(1))
        mpc._estimator_opts = dict(arrival=self.arrival_cost.nodes()[0],
                                   xprev_sym=self._x_prev.cat.nodes() + self._p_est_prev.cat.nodes(),
                                   lterm_end=not discrete, nl_dup=True)
        mpc.setup()
        self._mpc = mpc
        self.S = mpc.S
        ps = self._ps = mpc.structure
        M = ps.M
        # ---- the reference's layouts
        xs_l, zs_l, us_l = m._x.layout(), m._z.layout(), m._u.layout()
        self._eps_layout = mpc._eps_layout
        self.n_eps = ps.n_eps                                        # (N, or 1 with nl_cons_single_slack)
        self._opt_x_layout = Layout([
            Entry("_x", struct=xs_l, repeat=[N + 1, 1 + M]),
            Entry("_z", struct=zs_l, repeat=[N, max(M, 1)]),
            Entry("_u", struct=us_l, repeat=[N]),
            Entry("_w", struct=m._w.layout(), repeat=[N]),
            Entry("_v", struct=m._v.layout(), repeat=[N]),
            Entry("_eps", struct=self._eps_layout, repeat=[self.n_eps]),
            Entry("_p_est", struct=self._p_est.layout()),
        ])
        self._opt_p_layout = Layout([
            Entry("_x_prev", struct=xs_l),
            Entry("_p_est_prev", struct=self._p_est.layout()),
            Entry("_p_set", struct=self._p_set.layout()),
            Entry("_tvp", struct=m._tvp.layout(), repeat=N),
            Entry("_y_meas", struct=m._y.layout(), repeat=N),
        ])
        self.n_opt_x, self.n_opt_p = self._opt_x_layout.size, self._opt_p_layout.size
        self._opt_x_num = NumStruct(self._opt_x_layout, 0.0)
        self.opt_x_num_unscaled = NumStruct(self._opt_x_layout, 0.0)
        self._opt_p_num = NumStruct(self._opt_p_layout, 0.0)
        self.opt_x_scaling = NumStruct(self._opt_x_layout, 1.0)
        self.opt_x_scaling["_x"] = self._x_scaling.master
        self.opt_x_scaling["_u"] = self._u_scaling.master
        if nz:
            self.opt_x_scaling["_z"] = self._z_scaling.master
        if npe:
            self.opt_x_scaling["_p_est"] = self._p_est_scaling.master
        n_rows = ps.ne                                               # nl_cons rows of a stage (all evaluations)
        self._rows_stage = M * nx + max(M, 1) * nz + nx + ny + n_rows       # (discrete: M = 0 - algebraic rows, the rows x+ = f of the reference, then measurement / nl_cons rows)
        self.n_opt_lagr = N * self._rows_stage
        self.lam_g_num = np.zeros(self.n_opt_lagr)
        if not discrete:
            # chain rows of a stage that exist in the reference (the riding parameters' rows do not): interval function per finite
            # element [alg rows of point 0 | per collocation point: collocation rows, alg rows | end-of-element rows], continuity rows
            deg, ni, nxa = s.collocation_deg, s.collocation_ni, ps.nx
            st_rows = np.arange(nxa) < nx
            el = np.concatenate([np.ones(nz, bool)] + [np.concatenate([st_rows, np.ones(nz, bool)])] * deg + [st_rows])
            self._dyn_keep = np.concatenate([el] * ni + [st_rows])
            assert int(self._dyn_keep.sum()) == M * (nx + nz) + nx and self._dyn_keep.size == M * (nxa + nz) + nxa
        # offsets inside the reference's opt_x
        self._o_z = (N + 1) * (M + 1) * nx                          # end of `_x` = start of `_z` (N x max(M, 1) slots)
        self._o_u = self._o_z + N * max(M, 1) * nz
        self._o_w = self._o_u + N * nu
        self._o_v = self._o_w + N * nw
        self._o_eps = self._o_v + N * nv
        self._o_p = self._o_eps + self.n_eps * self._eps_layout.size
        self._po_pset = nx + npe
        self._po_tvp = self._po_pset + self.n_p_set
        self._po_y = self._po_tvp + N * m.n_tvp
        # numeric helpers: measurement noise and its cost gradient at the solution
        args = [am._x.cat, am._u.cat, m._z.cat, am._tvp.cat, am._p.cat]      # (z: the model's algebraic states of the LAST point of the interval)
        if discrete:            # (h reads the next state through its algebraic copy: evaluate with the next NODE state in its place)
            v_of_x = sym.substitute(v_of, z_next, m._x.cat)
            h0_x = sym.substitute(h0, z_next, m._x.cat)
        else:
            v_of_x, h0_x = v_of, h0
        self._v_fun = sym.Function("v_of", args, [v_of_x])
        dl_dv = sym.jacobian(self.stage_cost, m._v.cat).T if nv else sym.SX(np.zeros((0, 1)))
        self._dldv_fun = sym.Function("dldv", [m._w.cat, m._v.cat, m._tvp.cat, m._p.cat], [dl_dv])
        # (the reference's measurement rows read the NODE state x_{k+1}, here the stage cost reads the end slot of the interval:
        #  the multipliers of the continuity rows differ by (dh/dx)' lambda_meas)
        lam_y = sym.SX.sym("lam_y", ny, 1)
        self._hx_fun = sym.Function("hx_lam", args + [lam_y], [sym.jacobian(h0_x, m._x.cat).T @ lam_y])
        if self._y_free.size:
            # d/d(y of the free rows) of the chain problem's stage terms: collocation rows  h f(x_j, u, tvp, p) / s_x  (weighted by their
            # multipliers) and the stage cost at the end state - what the multipliers of the rows  u - y = 0  balance
            y_free = y_sym[[int(i) for i in self._y_free]]
            lam_c = sym.SX.sym("lam_c", ps.nx, 1)
            self._dfdy_fun = sym.Function("dfdy_lam", args + [lam_c], [sym.jacobian(am._rhs, y_free).T @ lam_c])
            self._dldy_fun = sym.Function("dldy", args, [sym.jacobian(stage, y_free).T])
        self._update_bounds()
        meta = {k: getattr(s, k) for k in ("n_horizon", "t_step", "meas_from_data", "state_discretization", "collocation_type",
                                           "collocation_deg", "collocation_ni", "nl_cons_check_colloc_points", "store_full_solution",
                                           "store_lagr_multiplier")}
        self.data.set_meta(**meta)
        self.flags["setup"] = True

    opt_x_num = property(lambda self: self._opt_x_num)
    opt_p_num = property(lambda self: self._opt_p_num)

    def _update_bounds(self):
        """_mhe.py:995-1028 on the chain problem: state bounds on every stored point (cons_check_colloc_points) or on the
        states `_x[1:N, -1]`; the bounds of `_p_est` on ONE of its copies (they are all equal at a feasible point)."""
        mpc, ps, s = self._mpc, self._ps, self.settings
        nx, nu, N, M = self.model.n_x, self.model.n_u, s.n_horizon, ps.M
        NXA, NUA = ps.nx, ps.nu
        lb, ub = mpc._lb_opt_x.master, mpc._ub_opt_x.master
        lb[:], ub[:] = -np.inf, np.inf
        XL, XU = lb[:ps.off_z].reshape(N + 1, M + 1, NXA), ub[:ps.off_z].reshape(N + 1, M + 1, NXA)
        xl, xu = self._x_lb.master / self._x_scaling.master, self._x_ub.master / self._x_scaling.master
        if s.cons_check_colloc_points:
            XL[:, :, :nx], XU[:, :, :nx] = xl, xu
        else:
            XL[1:N, -1, :nx], XU[1:N, -1, :nx] = xl, xu
        XL[0, -1, nx:], XU[0, -1, nx:] = self._p_est_lb.master / self._p_est_scaling.master, self._p_est_ub.master / self._p_est_scaling.master
        UL, UU = lb[ps.off_u:ps.off_eps].reshape(N, NUA), ub[ps.off_u:ps.off_eps].reshape(N, NUA)
        kp = self._u_keep
        UL[:, :kp.size], UU[:, :kp.size] = (self._u_lb.master / self._u_scaling.master)[kp], (self._u_ub.master / self._u_scaling.master)[kp]
        if self.model.n_z:        # (_mhe.py:1006-1007 every `_z` slot, :1015-1016 the first slot of every interval)
            nz = self.model.n_z
            if self._discrete:    # (one slot per interval: the model's algebraic states, then the copy of the next state)
                ZL, ZU = lb[ps.off_z:ps.off_u].reshape(N, 1, nz + nx)[:, :, :nz], ub[ps.off_z:ps.off_u].reshape(N, 1, nz + nx)[:, :, :nz]
            else:
                ZL, ZU = lb[ps.off_z:ps.off_u].reshape(N, -1, nz), ub[ps.off_z:ps.off_u].reshape(N, -1, nz)
            sl = slice(None) if s.cons_check_colloc_points else slice(0, 1)
            ZL[:, sl, :], ZU[:, sl, :] = self._z_lb.master / self._z_scaling.master, self._z_ub.master / self._z_scaling.master
        if mpc.n_eps:
            lb[ps.off_eps:].reshape(-1, mpc.n_eps)[:] = mpc._eps_lb
            ub[ps.off_eps:].reshape(-1, mpc.n_eps)[:] = mpc._eps_ub

    # ------------------------------------------------------------------ layouts: reference <-> chain problem
    def _to_chain(self, ox: np.ndarray) -> np.ndarray:
        """reference layout -> chain problem (a leading batch axis is carried through)"""
        ps, m, N = self._ps, self.model, self.settings.n_horizon
        nx, nu, nw, M = m.n_x, m.n_u, m.n_w, ps.M
        ox = np.asarray(ox, dtype=float)
        lead = ox.shape[:-1]
        out = np.zeros(lead + (ps.n_opt_x,))
        X = out[..., :ps.off_z].reshape(lead + (N + 1, M + 1, ps.nx))
        X[..., :nx] = ox[..., :self._o_z].reshape(lead + (N + 1, M + 1, nx))
        X[..., nx:] = ox[..., None, None, self._o_p:]
        U = out[..., ps.off_u:ps.off_eps].reshape(lead + (N, ps.nu))
        nuk = self._nuk                                             # (inputs measured without noise are no variables of the chain problem)
        U[..., :nuk] = ox[..., self._o_u:self._o_w].reshape(lead + (N, nu))[..., self._u_keep]
        if nw:
            U[..., nuk:] = ox[..., self._o_w:self._o_v].reshape(lead + (N, nw))
        out[..., ps.off_eps:] = ox[..., self._o_eps:self._o_p]
        if self._discrete:      # (the algebraic copy of the next state starts at the guess of that state)
            Z = out[..., ps.off_z:ps.off_u].reshape(lead + (N, m.n_z + nx))
            Z[..., m.n_z:] = X[..., 1:, -1, :nx]
            if m.n_z:
                Z[..., :m.n_z] = ox[..., self._o_z:self._o_u].reshape(lead + (N, m.n_z))
        elif m.n_z:             # (same block in both layouts: N x M slots, _mhe.py:1056 / _mpc.py:1130 with one scenario)
            out[..., ps.off_z:ps.off_u] = ox[..., self._o_z:self._o_u]
        return out

    def _z_last(self, cx: np.ndarray, lead: tuple) -> np.ndarray:
        """(unscaled) algebraic states of the last point of every interval, shape lead + (N, n_z) - what the measurement function reads"""
        ps, m, N = self._ps, self.model, self.settings.n_horizon
        if not m.n_z:
            return np.zeros(lead + (N, 0))
        if self._discrete:
            return cx[..., ps.off_z:ps.off_u].reshape(lead + (N, m.n_z + m.n_x))[..., :m.n_z] * self._z_scaling.master
        Z = cx[..., ps.off_z:ps.off_u].reshape(lead + (N, max(ps.M, 1), m.n_z))
        return Z[..., -1, :] * self._z_scaling.master

    def _p_to_chain(self, op: np.ndarray) -> np.ndarray:
        """opt_p: [_x_prev | _p_est_prev | _p_set | _tvp (N) | _y_meas (N)] -> [_x0 = previous estimate of (x, p_est) |
        _tvp = (tvp_k, y_k), N + 1 stages | _p = p_set | _u_prev = 0]"""
        ps, m, N = self._ps, self.model, self.settings.n_horizon
        op = np.asarray(op, dtype=float)
        lead = op.shape[:-1]
        P = np.zeros(lead + (ps.n_opt_p,))
        P[..., :ps.nx] = op[..., :ps.nx]
        TV = P[..., ps.p_off_tvp:ps.p_off_p].reshape(lead + (N + 1, ps.ntvp))
        if m.n_tvp:
            TV[..., :N, :m.n_tvp] = op[..., self._po_tvp:self._po_y].reshape(lead + (N, m.n_tvp))
        TV[..., :N, m.n_tvp:] = op[..., self._po_y:].reshape(lead + (N, m.n_y))
        P[..., ps.p_off_p:ps.p_off_uprev] = op[..., self._po_pset:self._po_tvp]
        return P

    def _from_chain(self, cx: np.ndarray, opt_p_chain: np.ndarray) -> np.ndarray:
        ps, m, N = self._ps, self.model, self.settings.n_horizon
        nx, nu, nw, nv, M = m.n_x, m.n_u, m.n_w, m.n_v, ps.M
        cx, opt_p_chain = np.asarray(cx, dtype=float), np.asarray(opt_p_chain, dtype=float)
        lead = cx.shape[:-1]
        out = np.zeros(lead + (self.n_opt_x,))
        X = cx[..., :ps.off_z].reshape(lead + (N + 1, M + 1, ps.nx))
        out[..., :self._o_z] = X[..., :nx].reshape(lead + (-1,))
        if m.n_z and not self._discrete:
            out[..., self._o_z:self._o_u] = cx[..., ps.off_z:ps.off_u]
        elif m.n_z:
            out[..., self._o_z:self._o_u] = cx[..., ps.off_z:ps.off_u].reshape(lead + (N, m.n_z + nx))[..., :m.n_z].reshape(lead + (-1,))
        U = cx[..., ps.off_u:ps.off_eps].reshape(lead + (N, ps.nu))
        nuk = self._nuk
        Uo = np.zeros(lead + (N, nu))
        Uo[..., self._u_keep] = U[..., :nuk]
        TV = opt_p_chain[..., ps.p_off_tvp:ps.p_off_p].reshape(lead + (N + 1, ps.ntvp))[..., :N, :]
        if self._u_fixed.size:      # an input measured without noise IS its measurement (scaled like every entry of opt_x)
            Uo[..., self._u_fixed] = TV[..., m.n_tvp + self._y_free] / self._u_scaling.master[self._u_fixed]
        out[..., self._o_u:self._o_w] = Uo.reshape(lead + (-1,))
        if nw:
            out[..., self._o_w:self._o_v] = U[..., nuk:].reshape(lead + (-1,))
        if nv:      # measurement noise of stage k from the end state of its interval
            Pm = np.broadcast_to(opt_p_chain[..., None, ps.p_off_p:ps.p_off_uprev], lead + (N, ps.np_))
            sx = self._sx_aug
            su = np.concatenate([self._u_scaling.master[self._u_keep], np.ones(ps.nu - nuk)])
            cols = lambda a: np.moveaxis(a, -1, 0).reshape(a.shape[-1], int(np.prod(a.shape[:-1])))        # noqa: E731   (numel, batch * N)
            V = np.asarray(self._v_fun.eval(cols(X[..., 1:, -1, :] * sx), cols(U * su), cols(self._z_last(cx, lead)), cols(TV), cols(Pm))[0])
            out[..., self._o_v:self._o_eps] = np.moveaxis(V.reshape((nv,) + lead + (N,)), 0, -1).reshape(lead + (-1,))
        out[..., self._o_eps:self._o_p] = cx[..., ps.off_eps:]
        out[..., self._o_p:] = X[..., 0, -1, nx:]
        return out

    def _lam_from_chain(self, lam_chain: np.ndarray, ox: np.ndarray, opt_p: NumStruct) -> np.ndarray:
        """multipliers in the reference's row order; the measurement rows h + v - y = 0 carry -d(stage cost)/dv"""
        ps, m, N = self._ps, self.model, self.settings.n_horizon
        nx, nw, nv, ny, M = m.n_x, m.n_w, m.n_v, m.n_y, ps.M
        out = np.zeros(self.n_opt_lagr).reshape(N, self._rows_stage)
        L = lam_chain[ps.nx:].reshape(N, -1)                         # (the chain problem keeps nx dummy initial rows)
        if self._discrete:
            # chain problem: rows z - f = 0 (multiplier la), then x+ - ... ; the reference's rows f - x+ = 0 carry -la
            nz = m.n_z
            out[:, :nz] = L[:, :nz]                                  # (the model's own algebraic rows: the same rows)
            out[:, nz:nz + nx] = -L[:, nz:nz + nx]
            n_dyn = nz + nx + ps.nx
        else:
            # rows of the interval function (optimizer.py:905-983: algebraic rows of point 0, per collocation point the collocation rows of
            # the states and the algebraic rows, the end-of-element rows), then the continuity rows: the rows of the riding parameters drop out
            keep = self._dyn_keep
            n_dyn = keep.size
            out[:, :int(keep.sum())] = L[:, :n_dyn][:, keep]
        if nv:
            W = ox[self._o_w:self._o_v].reshape(N, nw) if nw else np.zeros((N, 0))
            V = ox[self._o_v:self._o_eps].reshape(N, nv)
            TV = opt_p.master[self._po_tvp:self._po_y].reshape(N, m.n_tvp)
            pm = np.zeros(m.n_p)
            off = 0
            for n in m._p.names:
                k = m._p.vars[n].numel()
                if k:
                    src = ox[self._o_p:] * self._p_est_scaling.master if n in self._p_est.names else opt_p.master[self._po_pset:self._po_tvp]
                    grp = self._p_est if n in self._p_est.names else self._p_set
                    pm[off:off + k] = src[grp.offset(n):grp.offset(n) + k]
                off += k
            g = self._dldv_fun.eval(W.T, V.T, TV.T, np.tile(pm[:, None], (1, N)))[0]
            lam_meas = -np.asarray(g).reshape(nv, N).T
            r_meas = self._rows_stage - ny - (L.shape[1] - n_dyn)       # (first measurement row of a stage)
            lam_all = np.zeros((N, ny))
            lam_all[:, self._y_noisy] = lam_meas
            cx, Pc = self._mpc.opt_x_num.master, self._mpc.opt_p_num.master
            if self._y_free.size:
                lam_all[:, self._y_free] = self._fixed_input_multipliers(L, cx, Pc)
            out[:, r_meas:r_meas + ny] = lam_all
            X = cx[:ps.off_z].reshape(N + 1, M + 1, ps.nx)[1:, -1, :] * self._sx_aug
            U = cx[ps.off_u:ps.off_eps].reshape(N, ps.nu) * np.concatenate([self._u_scaling.master[self._u_keep], np.ones(ps.nu - self._nuk)])
            TVc = Pc[ps.p_off_tvp:ps.p_off_p].reshape(N + 1, ps.ntvp)[:N]
            Zl = self._z_last(cx, ())
            hx = self._hx_fun.eval(X.T, U.T, Zl.T, TVc.T, np.tile(Pc[ps.p_off_p:ps.p_off_uprev][:, None], (1, N)), lam_all.T)[0]
            if not self._discrete:
                out[:, r_meas - nx:r_meas] += np.asarray(hx).reshape(nx, N).T
        else:
            r_meas = self._rows_stage - ny - (L.shape[1] - n_dyn)
        out[:, r_meas + ny:] = L[:, n_dyn:]
        return out.reshape(-1)

    def _fixed_input_multipliers(self, L: np.ndarray, cx: np.ndarray, Pc: np.ndarray) -> np.ndarray:
        """Multipliers of the rows  u_j - y = 0  of the inputs measured without noise (reference rows `yk_calc - y_meas`,
        _mhe.py:1144-1158), shape (N, n_free).  In the chain problem such an input is the parameter y of its stage; stationarity of the
        reference's NLP w.r.t. the (scaled) input reads  s_u dL'/du + s_u nu = 0  with L' = every other term of the Lagrangian, and
        L' as a function of the physical input is the chain problem's Lagrangian as a function of that parameter:  nu = -dL_c/dy.
        The terms that depend on it: the collocation rows  h f(x_j, u, ..) / s_x - sum_r C[r, j] x_r  (multipliers L) and the stage
        cost (through the noise of measurements that read the input, if any).  Continuous models without algebraic states."""
        ps, m, s = self._ps, self.model, self.settings
        N, M, deg, ni = s.n_horizon, ps.M, s.collocation_deg, s.collocation_ni
        h = s.t_step / ni
        Xs = cx[:ps.off_z].reshape(N + 1, M + 1, ps.nx) * self._sx_aug
        U = cx[ps.off_u:ps.off_eps].reshape(N, ps.nu) * np.concatenate([self._u_scaling.master[self._u_keep], np.ones(ps.nu - self._nuk)])
        TVc = Pc[ps.p_off_tvp:ps.p_off_p].reshape(N + 1, ps.ntvp)[:N]
        Pm = np.tile(Pc[ps.p_off_p:ps.p_off_uprev][:, None], (1, N))
        Z0 = np.zeros((0, N))
        dL = np.asarray(self._dldy_fun.eval(Xs[1:, -1, :].T, U.T, Z0, TVc.T, Pm)[0], float).reshape(self._y_free.size, N)
        for i in range(ni):                       # rows of a stage: per finite element deg collocation blocks, then its end-of-element rows
            for j in range(1, deg + 1):
                slot = i * (deg + 1) + j - 1      # stored slot of collocation point j of element i (optimizer.py:905-935)
                r0 = (i * (deg + 1) + (j - 1)) * ps.nx
                lam = L[:, r0:r0 + ps.nx] * (h / self._sx_aug)
                dL = dL + np.asarray(self._dfdy_fun.eval(Xs[1:, slot, :].T, U.T, Z0, TVc.T, Pm, lam.T)[0], float).reshape(self._y_free.size, N)
        return -dL.T

    # ------------------------------------------------------------------ runtime
    def set_initial_guess(self) -> None:
        assert self.flags["setup"] is True, "mhe was not setup yet. Please call mhe.setup()."
        self._opt_x_num["_x"] = self._x0.master / self._x_scaling.master
        self._opt_x_num["_u"] = self._u0.master / self._u_scaling.master
        if self.model.n_z:
            self._opt_x_num["_z"] = self._z0.master / self._z_scaling.master
        if self.n_p_est:
            self._opt_x_num["_p_est"] = self._p_est0.master / self._p_est_scaling.master
        self.flags["set_initial_guess"] = True

    def solve(self) -> None:
        """the NLP of the current opt_p_num from the initial guess opt_x_num (Optimizer.solve, optimizer.py:731-787)"""
        mpc, ps, m, N = self._mpc, self._ps, self.model, self.settings.n_horizon
        P = mpc.opt_p_num.master
        P[:] = self._p_to_chain(self._opt_p_num.master)
        mpc.opt_x_num.master[:] = self._to_chain(self._opt_x_num.master)
        mpc.solve()
        self.solver_stats = mpc.solver_stats
        ox = self._from_chain(mpc.opt_x_num.master, P)
        self._opt_x_num.master[:] = ox
        self.opt_x_num_unscaled.master[:] = ox * self.opt_x_scaling.master
        self.lam_g_num = self._lam_from_chain(mpc.lam_g_num, ox, self._opt_p_num)

    def solve_batch(self, OPT_P: np.ndarray, OPT_X_INIT: np.ndarray) -> dict:
        """B estimation problems of this estimator (same model / settings, different parameter vectors: previous estimates,
        measurement windows, ...) in ONE device call - what a bank of estimators, or an estimator inside a batched closed loop,
        needs.  OPT_P: (B, n_opt_p), OPT_X_INIT: (B, n_opt_x), both in the reference's layouts; returns the solutions in the same
        layout, the estimates x(t_N) and p_est, and the solver statistics.  Needs `settings.max_batch >= B` at setup."""
        mpc, m = self._mpc, self.model
        OPT_P = np.asarray(OPT_P, dtype=float).reshape(-1, self.n_opt_p)
        B = OPT_P.shape[0]
        P = self._p_to_chain(OPT_P)
        Xi = self._to_chain(np.asarray(OPT_X_INIT, dtype=float).reshape(B, self.n_opt_x))
        r = self.S.solve_batch(Xi, mpc._lb_opt_x.master, mpc._ub_opt_x.master, mpc._nlp_cons_lb, mpc._nlp_cons_ub, P)
        ox = self._from_chain(r["x"], P)
        xN = ox[:, self._o_z - m.n_x:self._o_z] * self._x_scaling.master
        return {"opt_x": ox, "x": xN, "p_est": ox[:, self._o_p:] * self._p_est_scaling.master, "stats": r["stats"]}

    def make_step(self, y0: np.ndarray) -> np.ndarray:
        """_mhe.py:896-993: the current measurement in, the state estimate at the end of the horizon out"""
        assert self.flags["setup"] is True, "optimizer was not setup yet. Please call optimizer.setup()."
        y0 = np.asarray(y0.master if hasattr(y0, "master") else y0, dtype=float).reshape(-1)
        assert y0.size == self.model.n_y, "Wrong input with shape {}. Expected vector with {} elements".format(y0.shape, self.model.n_y)
        self.data.update(_y=y0)
        t0 = float(self._t0[0])
        x0, p_est0 = self._x0.master.copy(), self._p_est0.master.copy()
        tvp0, p_set0, y_traj = self.tvp_fun(t0), self.p_fun(t0), self.y_fun(t0)
        self._opt_p_num["_x_prev"] = _arr(self._opt_x_num["_x", 1, -1]) * self._x_scaling.master
        nx_, npe_ = self.model.n_x, self.n_p_est
        Pm = self._opt_p_num.master
        Pm[nx_:nx_ + npe_] = p_est0
        Pm[self._po_pset:self._po_tvp] = p_set0.master
        Pm[self._po_tvp:self._po_y] = tvp0.master
        Pm[self._po_y:] = y_traj.master
        self.solve()
        x_next = _arr(self._opt_x_num["_x", -1, -1]) * self._x_scaling.master
        p_est_next = _arr(self._opt_x_num["_p_est"]) * self._p_est_scaling.master if self.n_p_est else np.zeros(0)
        u0 = _arr(self._opt_x_num["_u", -1]) * self._u_scaling.master
        z0 = _arr(self._opt_x_num["_z", -1, -1]) * self._z_scaling.master if self.model.n_z else np.zeros(0)
        p0 = np.zeros(self.model.n_p)
        off = 0
        for n in self.model._p.names:
            k = self.model._p.vars[n].numel()
            if k:
                grp, src = (self._p_est, p_est0) if n in self._p_est.names else (self._p_set, p_set0.master)
                p0[off:off + k] = src[grp.offset(n):grp.offset(n) + k]
            off += k
        self.data.update(_x=x0, _u=u0, _p=p0, _time=self._t0)
        # auxiliary expressions of the LAST stage of the window (_mhe.py:959, 969, 1193-1195: aux(x[k, -1], u[k], z[k, -1], tvp[k], p)
        # with the parameters of the current solution) - ADVICE r3
        if self.model.n_aux:
            mdl = self.model
            xk = _arr(self._opt_x_num["_x", -2, -1]) * self._x_scaling.master
            pk = np.zeros(mdl.n_p)
            off = 0
            for n in mdl._p.names:
                k = mdl._p.vars[n].numel()
                if k:
                    grp, src = (self._p_est, p_est_next) if n in self._p_est.names else (self._p_set, p_set0.master)
                    pk[off:off + k] = src[grp.offset(n):grp.offset(n) + k]
                off += k
            tvk = _arr(tvp0["_tvp", -1]) if mdl.n_tvp else np.zeros(0)
            aux0 = mdl._aux_expression_fun.eval(xk.reshape(-1, 1), u0.reshape(-1, 1), z0.reshape(-1, 1), np.asarray(tvk, float).reshape(-1, 1),
                                                pk.reshape(-1, 1))[0]
            self.data.update(_aux=np.asarray(aux0, float).reshape(-1))
        if self.model.n_z:
            self.data.update(_z=z0)
        if self.model.n_tvp:
            self.data.update(_tvp=_arr(tvp0["_tvp", -1]))
        self.data.update(opt_p_num=self._opt_p_num.master.copy())
        if self.settings.store_full_solution:
            self.data.update(_opt_x_num=self.opt_x_num_unscaled.master.copy())
        if self.settings.store_lagr_multiplier:
            self.data.update(_lam_g_num=self.lam_g_num.copy())
        for k in self.settings.store_solver_stats:
            if k in self.solver_stats:
                self.data.update(**{k: np.array([float(self.solver_stats[k])])})
        self._t0 = self._t0 + self.settings.t_step
        self._x0.master[:] = x_next
        if self.n_p_est:
            self._p_est0.master[:] = p_est_next
        self._u0.master[:] = u0
        if self.model.n_z:
            self._z0.master[:] = z0
        return x_next.reshape(-1, 1)
