"""MPC controller with do_mpc.controller.MPC's user surface on the MI355X IPM backend.

Reference surface mirrored here (same names, argument meaning and error behaviour):
  MPCSettings                     /root/reference/do_mpc/controller/_controllersettings.py:63-175
  MPC.bounds / scaling / terminal_bounds   /root/reference/do_mpc/optimizer.py:268-439, _mpc.py:407-480
  MPC.set_objective / set_rterm   /root/reference/do_mpc/controller/_mpc.py:525-677
  MPC.set_nl_cons                 /root/reference/do_mpc/optimizer.py:483-541
  MPC.get_p_template / set_p_fun / set_uncertainty_values / tvp   _mpc.py:711-881, optimizer.py:588-676
  MPC.setup / set_initial_guess / make_step                       _mpc.py:933-1059
  Optimizer.solve                 /root/reference/do_mpc/optimizer.py:731-787
What differs is below the surface: setup() does not build one big symbolic NLP; it builds the
integer structure (do_mpc_amd/structure.py), lowers the model functions to a gfx950 code object
(do_mpc_amd/lowering.py, build.py) and creates a HipIpmSolver, which is stored in `self.S` exactly
where the reference stores the nlpsol object.
"""
from __future__ import annotations

import itertools
import time
import warnings
from dataclasses import asdict, dataclass, field
from typing import Callable, Dict, List, Optional, Union

import numpy as np

from . import lowering, nlp_route, sym
from .model import Model, VarGroup
from .solver import HipIpmSolver
from .structs import Entry, Layout, NumStruct
from .structure import build_structure, lagrange_collocation


# ----------------------------------------------------------------------------------------------
@dataclass
class MPCSettings:
    n_horizon: int = None
    t_step: float = None
    n_robust: int = 0
    open_loop: bool = False
    use_terminal_bounds: bool = False
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
    gpu_index: int = 0          # extension: which HIP device owns this controller
    max_batch: int = 1          # extension: capacity for make_step_batch
    block_threads: int = 0      # extension: threads per problem in batch mode (64/128/256; 0 = choose from max_batch)

    def check_for_mandatory_settings(self):
        if self.n_horizon is None:
            raise ValueError("n_horizon must be set")
        if self.t_step is None:
            raise ValueError("t_step must be set")

    def supress_ipopt_output(self):
        self.nlpsol_opts.update({"ipopt.print_level": 0, "ipopt.sb": "yes", "print_time": 0})

    def set_linear_solver(self, solver_name: str = "MA27"):
        self.nlpsol_opts.update({"ipopt.linear_solver": solver_name})   # accepted, meaningless here


class _Indexed:
    """`obj[...]` / `obj[...] = v` forwarding (the reference's IndexedProperty descriptor)."""

    def __init__(self, getter, setter):
        self._g, self._s = getter, setter

    def __getitem__(self, ind):
        return self._g(ind)

    def __setitem__(self, ind, val):
        self._s(ind, val)


class MPCData:
    """Per-step record store (subset of do_mpc.data.MPCData, /root/reference/do_mpc/data.py:173-218)."""

    def __init__(self, model: Model):
        self.model = model
        self._rows: Dict[str, list] = {}
        self.meta_data: Dict = {}

    def set_meta(self, **kw):
        self.meta_data.update(kw)

    def init_storage(self):
        """Drop all stored records (do_mpc.data.Data.init_storage, /root/reference/do_mpc/data.py:123-137)."""
        self._rows = {}

    def update(self, **kw):
        for k, v in kw.items():
            if hasattr(v, "master"):
                v = v.master
            elif hasattr(v, "arr"):
                v = v.arr
            self._rows.setdefault(k, []).append(np.array(v, dtype=float).reshape(-1))

    def __getitem__(self, key):
        if isinstance(key, tuple):
            name, rest = key[0], key[1:]
            arr = self[name]
            grp = self.model._getvar(name)
            off = grp.offset(rest[0])
            n = grp.vars[rest[0]].numel()
            return arr[:, off:off + n]
        rows = self._rows.get(key, [])
        return np.vstack(rows) if rows else np.zeros((0, 0))

    def prediction(self, ind: tuple, t_ind: int = -1) -> np.ndarray:
        """Predicted trajectory of one variable from the stored optimal solutions (do_mpc.data.MPCData.prediction,
        /root/reference/do_mpc/data.py:246-374): `data.prediction(('_x', 'T_R'))` -> array [n_size][n_horizon(+1)][n_scenario]
        of the solution stored at time index `t_ind`; the scenario axis follows the leaves of the tree through
        `structure_scenario` (the stage-k column of leaf j is its ancestor at stage k).  Needs `store_full_solution`."""
        assert isinstance(ind, tuple), "Query index must be of type tuple."
        lay_x, lay_p, lay_aux = self._layouts
        sc = np.asarray(self.meta_data["structure_scenario"], dtype=int)           # [N+1][n_leaves]
        rows = self._rows

        def stored(key, size):
            return np.vstack(rows[key])[t_ind] if rows.get(key) else np.zeros(size)

        kind = ind[0]
        if kind in ("_x", "_z"):
            idx = lay_x.resolve((kind, slice(None), slice(None), -1) + tuple(ind[1:]))        # [stage][scenario slot][element]
            cols = sc[:idx.shape[0], :]
            src = stored("_opt_x_num", lay_x.size)
        elif kind == "_u":
            idx = lay_x.resolve((kind, slice(None), slice(None)) + tuple(ind[1:]))
            cols = sc[:-1, [0]] if self.meta_data.get("open_loop") else sc[:-1, :]
            src = stored("_opt_x_num", lay_x.size)
        elif kind == "_aux":
            idx = lay_aux.resolve((kind, slice(None), slice(None)) + tuple(ind[1:]))
            cols = sc[:-1, :]
            src = stored("_opt_aux_num", lay_aux.size)
        elif kind == "_tvp":
            idx = lay_p.resolve((kind, slice(None)) + tuple(ind[1:]))
            return stored("opt_p_num", lay_p.size)[idx.reshape(1, -1, 1)]
        else:
            raise ValueError("Index {} not recognized.".format(kind))
        stage = np.arange(idx.shape[0])[:, None]
        return src[np.moveaxis(idx[stage, cols, :], -1, 0)]

    def __getattr__(self, key):
        if key.startswith("__"):
            raise AttributeError(key)
        rows = self.__dict__.get("_rows", {})
        if key in rows:
            return np.vstack(rows[key])
        model = self.__dict__.get("model")
        if model is not None and key in ("_x", "_u", "_z", "_y", "_p", "_tvp", "_aux"):      # nothing recorded yet: (0, n) like the reference
            return np.zeros((0, model._getvar(key).size))
        raise AttributeError(key)


# ----------------------------------------------------------------------------------------------
class MPC:
    def __init__(self, model: Model, settings: Optional[MPCSettings] = None):
        assert model.flags["setup"] is True, "Model for MPC was not setup. After the complete model creation call model.setup()."
        self.model = model
        self.settings = self._settings = settings if settings is not None else MPCSettings()
        m = model
        self._x_lb, self._x_ub = m._x(-np.inf), m._x(np.inf)
        self._u_lb, self._u_ub = m._u(-np.inf), m._u(np.inf)
        self._z_lb, self._z_ub = m._z(-np.inf), m._z(np.inf)
        self._x_terminal_lb, self._x_terminal_ub = m._x(-np.inf), m._x(np.inf)
        self._x_scaling, self._u_scaling = m._x(1.0), m._u(1.0)
        self._z_scaling, self._p_scaling = m._z(1.0), m._p(1.0)
        self.rterm_factor = m._u(0.0)
        # symbols of the previous input for user-defined rterm expressions (_mpc.py: `self.u_prev`, same names and shapes as _u)
        self.u_prev = VarGroup("_u_prev")
        for nm in m._u.names:
            self.u_prev.add(nm, sym.SX.sym("u_prev_" + nm, *m._u.vars[nm].shape))
        self.rterm_expr = None
        self._x0, self._u0, self._z0, self._t0 = m._x(0.0), m._u(0.0), m._z(0.0), np.array([0.0])
        self.nl_cons_list: List[dict] = []
        self.slack_vars_list: List[dict] = []
        self.slack_cost = 0
        self.n_combinations = 1
        self.flags = {"setup": False, "set_objective": False, "set_rterm": False, "set_tvp_fun": False,
                      "set_p_fun": False, "set_initial_guess": False, "prepare_nlp": False, "rterm_fun": False,
                      "MINLP": len(model.integer) > 0}
        self.bounds = _Indexed(self._get_bounds, self._set_bounds)
        self.scaling = _Indexed(self._get_scaling, self._set_scaling)
        self.terminal_bounds = _Indexed(self._get_terminal_bounds, self._set_terminal_bounds)
        self.data = MPCData(model)
        self.solver_stats: dict = {}
        self.S: Optional[HipIpmSolver] = None

    # ------------------------------------------------------------------ legacy settings interface
    def set_param(self, **kwargs) -> None:
        assert self.flags["setup"] is False, "Setting parameters after setup is prohibited."
        for key, value in kwargs.items():
            if hasattr(self.settings, key):
                setattr(self.settings, key, value)
            else:
                print("Warning: Key {} does not exist for MPC.".format(key))

    # ------------------------------------------------------------------ bounds / scaling
    def _bound_struct(self, ind, terminal=False):
        assert isinstance(ind, tuple), "Power index must include bound_type, var_type, var_name (as a tuple)."
        assert len(ind) >= 2, "Power index must include bound_type, var_type, var_name (as a tuple)."
        bound_type, var_type = ind[0], ind[1]
        var_name = ind[2:]
        if bound_type not in ("lower", "upper"):
            raise Exception("Invalid power index {} for bound_type. Must be from (lower, upper).".format(bound_type))
        if terminal:
            if var_type not in ("_x", "x"):
                raise Exception("Invalid power index {} for var_type. Must be _x.".format(var_type))
            tab = {"lower": self._x_terminal_lb, "upper": self._x_terminal_ub}
        else:
            if var_type not in ("_x", "_u", "_z"):
                raise Exception("Invalid power index {} for var_type. Must be from (_x, _u, _z).".format(var_type))
            tab = {("lower", "_x"): self._x_lb, ("upper", "_x"): self._x_ub, ("lower", "_u"): self._u_lb,
                   ("upper", "_u"): self._u_ub, ("lower", "_z"): self._z_lb, ("upper", "_z"): self._z_ub}
            return tab[(bound_type, var_type)], var_name
        return tab[bound_type], var_name

    def _get_bounds(self, ind):
        st, name = self._bound_struct(ind)
        return st[name] if name else st

    def _set_bounds(self, ind, val):
        st, name = self._bound_struct(ind)
        st[name] = val

    def _get_terminal_bounds(self, ind):
        st, name = self._bound_struct(ind, terminal=True)
        return st[name] if name else st

    def _set_terminal_bounds(self, ind, val):
        st, name = self._bound_struct(ind, terminal=True)
        st[name] = val

    def _scaling_struct(self, ind):
        assert isinstance(ind, tuple), "Power index must include var_type, var_name (as a tuple)."
        var_type = ind[0]
        if var_type not in ("_x", "_u", "_z", "_p"):
            raise Exception("Invalid power index {} for var_type. Must be from (_x, _u, _z, _p).".format(var_type))
        return {"_x": self._x_scaling, "_u": self._u_scaling, "_z": self._z_scaling, "_p": self._p_scaling}[var_type], ind[1:]

    def _get_scaling(self, ind):
        st, name = self._scaling_struct(ind)
        return st[name] if name else st

    def _set_scaling(self, ind, val):
        st, name = self._scaling_struct(ind)
        st[name] = val

    # ------------------------------------------------------------------ initial values
    def _set_iter(self, attr, val):
        st = getattr(self, attr)
        if hasattr(val, "master"):
            val = val.master
        elif hasattr(val, "arr"):
            val = val.arr
        val = np.asarray(val, dtype=float).reshape(-1)
        assert val.size == st.size, "Wrong input with shape {}. Expected vector with {} elements".format(val.shape, st.size)
        st.master[:] = val

    x0 = property(lambda self: self._x0, lambda self, v: self._set_iter("_x0", v))
    u0 = property(lambda self: self._u0, lambda self, v: self._set_iter("_u0", v))
    z0 = property(lambda self: self._z0, lambda self, v: self._set_iter("_z0", v))

    def reset_history(self) -> None:
        """All stored records are removed, t0 = 0 (/root/reference/do_mpc/optimizer.py:441-446)."""
        self.data.init_storage()
        self._t0 = np.array([0.0])

    @property
    def t0(self):
        return self._t0

    @t0.setter
    def t0(self, v):
        self._t0 = np.array(v, dtype=float).reshape(-1)[:1]

    # ------------------------------------------------------------------ objective / constraints
    def set_objective(self, mterm=None, lterm=None) -> None:
        assert self.flags["setup"] is False, "Cannot call .set_objective after .setup()."
        for nm, t in (("mterm", mterm), ("lterm", lterm)):
            if t is not None and not isinstance(t, (sym.SX, sym.DM, int, float)):
                raise Exception("{} must be a symbolic expression. You have: {}.".format(nm, type(t)))
        self.mterm = sym.SX(0.0) if mterm is None else sym.SX(mterm)
        self.lterm = sym.SX(0.0) if lterm is None else sym.SX(lterm)
        if self.mterm.shape != (1, 1):
            raise Exception("mterm must have shape=(1,1). You have {}".format(self.mterm.shape))
        if self.lterm.shape != (1, 1):
            raise Exception("lterm must have shape=(1,1). You have {}".format(self.lterm.shape))
        m = self.model
        bad = sym.depends_on(self.mterm.nodes(), m._u.cat.nodes() + m._z.cat.nodes())
        if bad:
            raise Exception("mterm contains invalid symbolic variables as inputs. Must contain only: _x, _tvp, _p")
        self.flags["set_objective"] = True

    def set_rterm(self, rterm=None, **kwargs) -> None:
        assert self.flags["setup"] is False, "Cannot call .set_rterm after .setup()."
        if rterm is not None:
            # user-defined penalty (_mpc.py:593-677): scalar expression in _x, _u, mpc.u_prev, _tvp, _p
            rterm = sym.SX(rterm)
            assert rterm.shape == (1, 1), "rterm must have shape=(1,1). You have {}".format(rterm.shape)
            self.rterm_expr = rterm
            self.flags["rterm_fun"] = True
            self.flags["set_rterm"] = True
            return
        for key, val in kwargs.items():
            assert key in self.model._u.keys(), \
                "Must pass keywords that refer to input names defined in model. Valid is: {}. You have: {}".format(self.model._u.keys(), key)
            assert isinstance(val, (int, float, np.ndarray)), \
                "Value for {} must be int, float or numpy.ndarray. You have: {}".format(key, type(val))
            self.rterm_factor[key] = val
        self.flags["set_rterm"] = True

    def set_nl_cons(self, expr_name: str, expr, ub: float = np.inf, soft_constraint: bool = False,
                    penalty_term_cons: float = 1, maximum_violation: float = np.inf):
        assert self.flags["setup"] is False, "Cannot call .set_expression after .setup()."
        assert isinstance(expr_name, str), "expr_name must be str, you have: {}".format(type(expr_name))
        assert isinstance(expr, sym.SX), "expr must be a symbolic expression, you have: {}".format(type(expr))
        assert isinstance(ub, (int, float, np.ndarray)), "ub must be float, int or numpy.ndarray, you have: {}".format(type(ub))
        assert isinstance(soft_constraint, bool), "soft_constraint must be boolean, you have: {}".format(type(soft_constraint))
        if soft_constraint:
            self.slack_vars_list.append({"slack_name": expr_name, "shape": expr.shape, "ub": maximum_violation,
                                         "penalty": penalty_term_cons})
        self.nl_cons_list.append({"expr_name": expr_name, "expr": expr, "ub": ub})
        return expr

    # ------------------------------------------------------------------ parameters
    def get_p_template(self, n_combinations: int) -> NumStruct:
        self.n_combinations = n_combinations
        return NumStruct(Layout([Entry("_p", struct=self.model._p.layout(), repeat=n_combinations)]), 0.0)

    def set_p_fun(self, p_fun: Callable) -> None:
        assert self.get_p_template(self.n_combinations).labels() == p_fun(0).labels(), \
            "Incorrect output of p_fun. Use get_p_template to obtain the required structure."
        self.flags["set_p_fun"] = True
        self.p_fun = p_fun

    def set_uncertainty_values(self, **kwargs) -> None:
        if not kwargs:
            return None
        names = list(kwargs.keys())
        valid_names = self.model.p.keys()
        err_msg = "You passed keywords {}. Valid keywords are: {} (refering to user-defined parameter names)."
        assert set(names).issubset(set(valid_names)), err_msg.format(names, valid_names)
        p_scenario = list(itertools.product(*kwargs.values()))
        p_template = self.get_p_template(len(p_scenario))
        for c, combo in enumerate(p_scenario):
            for nm, val in zip(names, combo):
                p_template["_p", c, nm] = val

        def p_fun(t_now):
            return p_template

        self.set_p_fun(p_fun)

    def get_tvp_template(self) -> NumStruct:
        return NumStruct(Layout([Entry("_tvp", struct=self.model._tvp.layout(), repeat=self.settings.n_horizon + 1)]), 0.0)

    def set_tvp_fun(self, tvp_fun: Callable) -> None:
        assert isinstance(tvp_fun(0), NumStruct), "Incorrect output of tvp_fun. Use get_tvp_template to obtain the required structure."
        assert self.get_tvp_template().labels() == tvp_fun(0).labels(), \
            "Incorrect output of tvp_fun. Use get_tvp_template to obtain the required structure."
        self.flags["set_tvp_fun"] = True
        self.tvp_fun = tvp_fun

    # ------------------------------------------------------------------ validity (_mpc.py:883-931)
    def _check_validity(self):
        if not self.flags["set_objective"]:
            raise Exception("Objective is undefined. Please call .set_objective() prior to .setup().")
        if not self.flags["set_rterm"]:
            warnings.warn("rterm was not set and defaults to zero. Changes in the control inputs are not penalized. "
                          "Can lead to oscillatory behavior.")
        if not self.flags["set_tvp_fun"] and self.model._tvp.size > 0:
            raise Exception("You have not supplied a function to obtain the time-varying parameters defined in model. "
                            "Use .set_tvp_fun() prior to setup.")
        if not self.flags["set_p_fun"] and self.model._p.size > 0:
            raise Exception("You have not supplied a function to obtain the parameters defined in model. Use "
                            ".set_p_fun() (low-level API) or .set_uncertainty_values() (high-level API) prior to setup.")
        if np.any(self.rterm_factor.master < 0):
            warnings.warn("You have selected negative values for the rterm penalizing changes in the control input.")
        for lb, ub in ((self._x_lb, self._x_ub), (self._u_lb, self._u_ub), (self._z_lb, self._z_ub)):
            bad = lb.master > ub.master
            if np.any(bad):
                fail = [lab for i, lab in enumerate(lb.labels()) if bad[i]]
                raise Exception("Your bounds are inconsistent. For {} you have lower bound > upper bound.".format(fail))
        if np.all(self._x_terminal_ub.master == np.inf) and self.settings.use_terminal_bounds:
            self._x_terminal_ub = self._x_ub
        if np.all(self._x_terminal_lb.master == -np.inf) and self.settings.use_terminal_bounds:
            self._x_terminal_lb = self._x_lb
        if "tvp_fun" not in self.__dict__:
            _tvp = self.get_tvp_template()
            self.set_tvp_fun(lambda t: _tvp)
        if "p_fun" not in self.__dict__:
            _p = self.get_p_template(1)
            self.set_p_fun(lambda t: _p)

    # ------------------------------------------------------------------ setup
    def setup(self) -> None:
        self.prepare_nlp()
        self.create_nlp()

    def prepare_nlp(self) -> None:
        s, m = self.settings, self.model
        s.check_for_mandatory_settings()
        if self.flags["MINLP"]:
            raise NotImplementedError("structured HIP backend: integer inputs (MINLP/bonmin) are not supported")
        # nl_cons_check_colloc_points (_mpc.py:1229-1237): the rows are evaluated at every stored point of the interval,
        # `_x[k+1, s, i]`, `_z[k, s, i]` with the PARENT's scenario index s - the edge's own unknowns only on a scenario chain
        self._nl_colloc = bool(s.nl_cons_check_colloc_points and self.nl_cons_list and m.model_type == "continuous")
        if s.nl_cons_check_colloc_points and self.nl_cons_list and m.model_type == "discrete":
            # (the reference loops over range(n_total_coll_points) = range(0) there: the rows silently disappear from its NLP)
            raise NotImplementedError("structured HIP backend: nl_cons_check_colloc_points with a discrete model (the reference "
                                      "drops the nl_cons rows in that combination); switch the setting off")
        if self._nl_colloc and s.n_robust > 0 and self.n_combinations > 1:
            raise NotImplementedError("structured HIP backend: nl_cons_check_colloc_points is supported for scenario chains "
                                      "(one parameter combination) only")
        if m.n_z and s.n_robust > 0 and self.n_combinations > 1:
            zn = m._z.cat.nodes()
            if sym.depends_on(self.lterm.nodes(), zn) or any(sym.depends_on(c["expr"].nodes(), zn) for c in self.nl_cons_list):
                # (_mpc.py:1213 vs 1241, 1252: the dynamics read `_z[k, child]`, stage cost and nl_cons `_z[k, s]` of the PARENT's
                #  scenario index - on a branching tree that couples an edge with the unknowns of a different edge)
                raise NotImplementedError("structured HIP backend: a stage cost / nl_cons that depends on algebraic states "
                                          "is supported for scenario chains (n_robust = 0) only")
        if s.state_discretization != "collocation" and m.model_type == "continuous":
            raise Exception("Unknown state_discretization: {}".format(s.state_discretization))
        # size limit of an edge (kernel: the collocation / algebraic unknowns of an interval are eliminated inside one wavefront's
        # registers / LDS region; csrc/dompc_edge.h asserts the same bound at compile time)
        pts = (s.collocation_deg + 1) * s.collocation_ni if m.model_type == "continuous" else 0
        n_w = pts * m.n_x + max(pts, 1) * m.n_z
        # 128 on the in-LDS elimination of intervals with several finite elements (round 5); 64 on the single-element path (one extended
        # column per lane, registers) and on the dense edge path of DAE models / rows at the collocation points (one row per lane)
        n_w_max = 128 if (m.n_z == 0 and not self._nl_colloc and s.collocation_ni >= 2) else 64
        if n_w > n_w_max:
            raise NotImplementedError("structured HIP backend: {} collocation / algebraic unknowns per control interval "
                                      "((deg + 1) * ni * n_x + points * n_z); the kernels eliminate at most {} per interval - "
                                      "lower collocation_deg / collocation_ni".format(n_w, n_w_max))
        self._check_validity()
        # slack / nl_cons bookkeeping (optimizer.py:543-585)
        eps_entries = [Entry(sl["slack_name"], sl["shape"]) for sl in self.slack_vars_list]
        self._eps_layout = Layout(eps_entries)
        self.n_eps = self._eps_layout.size
        # a vector-valued expression contributes one row per element (rows and slack elements in the reference's
        # order: vertcat of the expressions, `_eps` entries with the expression's shape, optimizer.py:560-585)
        slack_names = [sl["slack_name"] for sl in self.slack_vars_list]
        slack_off = np.concatenate([[0], np.cumsum([int(np.prod(sl["shape"])) for sl in self.slack_vars_list])]).astype(int)
        self._nl_rows, ub_rows, self._nl_slack_index = [], [], []
        for c in self.nl_cons_list:
            nodes = c["expr"].nodes()
            ub = np.broadcast_to(np.asarray(c["ub"], dtype=float).reshape(-1), (len(nodes),)) if np.size(c["ub"]) in (1, len(nodes)) \
                else None
            assert ub is not None, "ub of nl_cons '{}' does not match the shape of the expression".format(c["expr_name"])
            si = slack_names.index(c["expr_name"]) if c["expr_name"] in slack_names else -1
            for i, nd in enumerate(nodes):
                self._nl_rows.append(nd)
                ub_rows.append(float(ub[i]))
                self._nl_slack_index.append(int(slack_off[si]) + i if si >= 0 else -1)
        self._nl_cons_ub = np.array(ub_rows)
        self._nl_cons_lb = -np.inf * np.ones(len(self._nl_rows))

        def per_element(key):
            out = []
            for sl in self.slack_vars_list:
                n = int(np.prod(sl["shape"]))
                v = np.asarray(sl[key], dtype=float).reshape(-1)
                assert v.size in (1, n), "{} of slack '{}' does not match its shape".format(key, sl["slack_name"])
                out.extend(np.broadcast_to(v, (n,)) if v.size == 1 else v)
            return np.array(out, dtype=float)

        self._eps_lb = np.zeros(self.n_eps)
        self._eps_ub = per_element("ub") if self.n_eps else np.zeros(0)
        self._eps_pen = per_element("penalty") if self.n_eps else np.zeros(0)

        discrete = m.model_type == "discrete"
        n_eval = (s.collocation_deg + 1) * s.collocation_ni if self._nl_colloc else 1      # evaluations of the rows per edge
        est = getattr(self, "_estimator_opts", None) or {}          # (set by do_mpc_amd.estimator.MHE: free initial state etc.)
        if est.get("nl_dup") and self._nl_rows:
            n_eval += 1                                             # (_mhe.py:1186-1188: the rows of the last point once more)
        self._structure_args = dict(nx=m.n_x, nu=m.n_u, nz=m.n_z, np_=m.n_p, ntvp=m.n_tvp, ne=len(self._nl_rows) * n_eval,
                                    ns=self.n_eps, deg=s.collocation_deg, ni=s.collocation_ni, N=s.n_horizon,
                                    n_comb=self.n_combinations, n_robust=s.n_robust, discrete=discrete,
                                    open_loop=bool(s.open_loop), single_slack=bool(s.nl_cons_single_slack))
        ps = build_structure(**self._structure_args)
        self.structure = ps
        self.scenario_tree = ps.scenario_tree
        if ps.open_loop_stack:
            from . import open_loop
            open_loop.check_supported(self)
        xs_l, zs_l, us_l = m._x.layout(), m._z.layout(), m._u.layout()
        self._opt_x_layout = Layout([
            Entry("_x", struct=xs_l, repeat=[s.n_horizon + 1, ps.S, 1 + ps.M]),
            Entry("_z", struct=zs_l, repeat=[s.n_horizon, ps.S, max(ps.M, 1)]),
            Entry("_u", struct=us_l, repeat=[s.n_horizon, ps.SU]),
            Entry("_eps", struct=self._eps_layout, repeat=[ps.n_eps, ps.S]),
        ])
        self._opt_p_layout = Layout([
            Entry("_x0", struct=xs_l),
            Entry("_tvp", struct=m._tvp.layout(), repeat=s.n_horizon + 1),
            Entry("_p", struct=m._p.layout(), repeat=self.n_combinations),
            Entry("_u_prev", struct=us_l),
        ])
        self._opt_aux_layout = Layout([Entry("_aux", struct=m._aux.layout(), repeat=[s.n_horizon, ps.S])])
        assert self._opt_x_layout.size == ps.n_opt_x and self._opt_p_layout.size == ps.n_opt_p
        self.n_opt_x, self.n_opt_p, self.n_opt_aux = ps.n_opt_x, ps.n_opt_p, self._opt_aux_layout.size
        self.n_opt_lagr = ps.n_g
        self.opt_x_scaling = NumStruct(self._opt_x_layout, 1.0)
        self.opt_x_scaling["_x"] = self._x_scaling.master
        self.opt_x_scaling["_z"] = self._z_scaling.master
        self.opt_x_scaling["_u"] = self._u_scaling.master
        self._lb_opt_x = NumStruct(self._opt_x_layout, -np.inf)
        self._ub_opt_x = NumStruct(self._opt_x_layout, np.inf)
        self.lb_opt_x = _Indexed(lambda i: self._lb_opt_x[i], lambda i, v: self._set_opt_bound(self._lb_opt_x, i, v))
        self.ub_opt_x = _Indexed(lambda i: self._ub_opt_x[i], lambda i, v: self._set_opt_bound(self._ub_opt_x, i, v))
        self._update_bounds()
        g_lb, g_ub = np.zeros(ps.n_g), np.zeros(ps.n_g)
        if ps.ne:
            for e in range(ps.n_edges):
                r0 = ps.tables["edge_row0"][e] + ps.rows_block + ps.nx
                g_lb[r0:r0 + ps.ne] = np.tile(self._nl_cons_lb, n_eval)
                g_ub[r0:r0 + ps.ne] = np.tile(self._nl_cons_ub, n_eval)
        # the low-level route prepare_nlp -> modify -> create_nlp (optimizer.py:82-215): until create_nlp these are LISTS with one entry
        # per constraint block - here the structured block of n_g rows first, user additions behind it (nlp_route.py)
        self._nlp_obj = nlp_route.NlpObjective()
        self._nlp_cons = [nlp_route.StructuredBlock("constraints", ps.n_g)]
        self._nlp_cons_lb = [g_lb]
        self._nlp_cons_ub = [g_ub]
        self._opt_x_sym = self._opt_p_sym = self._opt_x_unscaled_sym = self._aux_struct = None
        self._opt_x_num = NumStruct(self._opt_x_layout, 0.0)
        self.opt_x_num_unscaled = NumStruct(self._opt_x_layout, 0.0)
        self._opt_p_num = NumStruct(self._opt_p_layout, 0.0)
        self.opt_aux_num = NumStruct(self._opt_aux_layout, 0.0)
        self.lam_g_num = np.zeros(ps.n_g)
        self.lam_x_num = np.zeros(ps.n_opt_x)
        self.opt_g_num = np.zeros(ps.n_g)
        self.flags["prepare_nlp"] = True

    opt_x_num = property(lambda self: self._opt_x_num)
    opt_p_num = property(lambda self: self._opt_p_num)

    # ---- symbolic side of the NLP (_mpc.py:323-405, optimizer.py:82-215): created on first use
    def _need_prepared(self):
        assert self.flags["prepare_nlp"], "Cannot query attribute prior to calling MPC.prepare_nlp or MPC.setup"

    @property
    def opt_x(self):
        """Symbolic struct of the optimisation variables in the reference's layout and (scaled) units: `opt_x['_x', k, s, i]`,
        `opt_x['_u', k, s]`, `opt_x['_z', k, s, i]`, `opt_x['_eps', k, s]` (_mpc.py:323-372)."""
        self._need_prepared()
        if self._opt_x_sym is None:
            self._opt_x_sym = nlp_route.OptSymStruct(self._opt_x_layout, "opt_x")
        return self._opt_x_sym

    @property
    def opt_p(self):
        """Symbolic struct of the NLP parameters `[_x0 | _tvp | _p | _u_prev]` (_mpc.py:375-405)."""
        self._need_prepared()
        if self._opt_p_sym is None:
            self._opt_p_sym = nlp_route.OptSymStruct(self._opt_p_layout, "opt_p")
        return self._opt_p_sym

    @property
    def opt_x_unscaled(self):
        """opt_x in physical units: `opt_x(opt_x.cat * opt_x_scaling)` (_mpc.py:1157); a list-of-vectors view like opt_x"""
        self._need_prepared()
        if self._opt_x_unscaled_sym is None:
            ox = self.opt_x
            st = nlp_route.OptSymStruct.__new__(nlp_route.OptSymStruct)
            st.layout, st.prefix = ox.layout, "opt_x_unscaled"
            st.vec = ox.vec * sym.SX(list(self.opt_x_scaling.master), (ox.layout.size, 1))
            st.index_of = {}
            self._opt_x_unscaled_sym = st
        return self._opt_x_unscaled_sym

    @property
    def aux_struct(self):
        """Symbolic struct of the auxiliary expressions over the horizon, `aux_struct['_aux', k, s]` (_mpc.py:1171-1175;
        data.py:463 stores it as `data.opt_aux`)."""
        self._need_prepared()
        if self._aux_struct is None:
            self._aux_struct = nlp_route.OptSymStruct(self._opt_aux_layout, "opt_aux")
        return self._aux_struct

    @property
    def nlp_obj(self):
        """Handle on the NLP objective (optimizer.py:82-117).  `mpc.nlp_obj += expr` records an added term; create_nlp() lowers it
        or refuses it by name (nlp_route.check_additions)."""
        self._need_prepared()
        return self._nlp_obj

    @nlp_obj.setter
    def nlp_obj(self, val):
        self._need_prepared()
        assert not self.flags["setup"], "Cannot change attribute after calling MPC.create_nlp or MPC.setup"
        self._nlp_obj = val

    @property
    def nlp_cons(self):
        """Before create_nlp(): list of constraint blocks - the structured block (n_g rows, reference order) first; append symbolic
        expressions over opt_x / opt_p together with nlp_cons_lb / nlp_cons_ub entries (optimizer.py:119-169).  Afterwards: the
        concatenation."""
        self._need_prepared()
        return self._nlp_cons

    @nlp_cons.setter
    def nlp_cons(self, val):
        self._need_prepared()
        assert not self.flags["setup"], "Cannot change attribute after calling create_nlp or setup"
        self._nlp_cons = val

    @property
    def nlp_cons_lb(self):
        """lower bounds matching nlp_cons: a list of arrays before create_nlp(), their concatenation afterwards (optimizer.py:172-192)"""
        self._need_prepared()
        return self._nlp_cons_lb

    @nlp_cons_lb.setter
    def nlp_cons_lb(self, val):
        self._need_prepared()
        self._nlp_cons_lb = val

    @property
    def nlp_cons_ub(self):
        """upper bounds matching nlp_cons (optimizer.py:194-215)"""
        self._need_prepared()
        return self._nlp_cons_ub

    @nlp_cons_ub.setter
    def nlp_cons_ub(self, val):
        self._need_prepared()
        self._nlp_cons_ub = val

    def _set_opt_bound(self, st: NumStruct, ind, val):
        st[ind] = val
        idx = st.layout.resolve(ind).reshape(-1)
        st.master[idx] = st.master[idx] / self.opt_x_scaling.master[idx]     # optimizer.py:239,265

    def _update_bounds(self):
        """_mpc.py:1061-1095"""
        s = self.settings
        N = s.n_horizon
        if s.cons_check_colloc_points:
            self.lb_opt_x["_x", 1:N] = self._x_lb.master
            self.ub_opt_x["_x", 1:N] = self._x_ub.master
            self.lb_opt_x["_z"] = self._z_lb.master
            self.ub_opt_x["_z"] = self._z_ub.master
        else:
            self.lb_opt_x["_x", 1:N, :, -1] = self._x_lb.master
            self.ub_opt_x["_x", 1:N, :, -1] = self._x_ub.master
            self.lb_opt_x["_z", :, :, 0] = self._z_lb.master
            self.ub_opt_x["_z", :, :, 0] = self._z_ub.master
        self.lb_opt_x["_x", N, :, -1] = self._x_terminal_lb.master
        self.ub_opt_x["_x", N, :, -1] = self._x_terminal_ub.master
        self.lb_opt_x["_u"] = self._u_lb.master
        self.ub_opt_x["_u"] = self._u_ub.master
        if self.n_eps:
            self.lb_opt_x["_eps"] = self._eps_lb
            self.ub_opt_x["_eps"] = self._eps_ub

    def _lower(self) -> str:
        s, m, ps = self.settings, self.model, self.structure
        discrete = ps.discrete
        if discrete:
            C, D, h = np.zeros((1, 1)), np.zeros(1), 1.0
        else:
            _, C, D = lagrange_collocation(s.collocation_deg, s.collocation_type)
            h = s.t_step / s.collocation_ni
        if m.n_w and sym.depends_on(m._rhs.nodes(), m._w.cat.nodes()):
            rhs = sym.substitute(m._rhs, m._w.cat, sym.SX.zeros(m.n_w, 1)).nodes()     # _w = 0 in the MPC (_mpc.py:1166)
        else:
            rhs = m._rhs.nodes()
        alg = m._alg.nodes() if m.n_z else []
        if m.n_z and m.n_w and sym.depends_on(alg, m._w.cat.nodes()):
            alg = sym.substitute(m._alg, m._w.cat, sym.SX.zeros(m.n_w, 1)).nodes()
        nl_exprs = list(self._nl_rows)
        return lowering.lower_model(
            nx=m.n_x, nu=m.n_u, np_=m.n_p, ntvp=m.n_tvp,
            x_sym=m._x.cat.nodes(), u_sym=m._u.cat.nodes(), tvp_sym=m._tvp.cat.nodes(), p_sym=m._p.cat.nodes(),
            rhs=rhs, lterm=self.lterm.nodes()[0], mterm=self.mterm.nodes()[0], nl_exprs=nl_exprs,
            nl_slack_index=self._nl_slack_index, eps_penalty=self._eps_pen,
            sx=self._x_scaling.master, su=self._u_scaling.master, rterm=self.rterm_factor.master,
            h_scale=h, deg=s.collocation_deg, ni=s.collocation_ni, discrete=discrete, C=C, D=D,
            name=type(m).__name__, nz=m.n_z, z_sym=m._z.cat.nodes(), alg=alg, sz=self._z_scaling.master,
            sp=self._p_scaling.master,
            rterm_expr=(self.rterm_expr.nodes()[0] if self.rterm_expr is not None else None),
            uprev_sym=self.u_prev.cat.nodes(), nl_colloc=self._nl_colloc, eps_global=ps.eps_global,
            extras=getattr(self, "_nlp_extras", None), rows=getattr(self, "_nlp_rows", None),
            **{k: v for k, v in (getattr(self, "_estimator_opts", None) or {}).items()
               if k in ("arrival", "xprev_sym", "lterm_end", "nl_dup")})

    def create_nlp(self, _solver_factory=None) -> None:
        assert self.flags["prepare_nlp"], "call prepare_nlp() first"
        if not self.flags["setup"]:
            # the low-level route (optimizer.py:1050-1094, _mpc.py:1303-1310): what the user added after prepare_nlp() is lowered or
            # refused by name; afterwards the attributes are the concatenations, as in the reference
            nlp_route.check_additions(self)
            self._nlp_cons_lb = np.ascontiguousarray(np.concatenate([np.asarray(b, dtype=float).reshape(-1) for b in self._nlp_cons_lb]))
            self._nlp_cons_ub = np.ascontiguousarray(np.concatenate([np.asarray(b, dtype=float).reshape(-1) for b in self._nlp_cons_ub]))
            # (with accepted node-local rows the concatenation is longer than the structured block: n_g + appended rows, the reference's order)
            self._nlp_cons = self._nlp_cons[0] if self._nlp_rows is None else \
                nlp_route.StructuredBlock("constraints", self._nlp_cons_lb.size)
        if self.structure.open_loop_stack:
            # open_loop with several scenarios: not tree-structured - a chain over the stacked scenario states, behind the same callable
            from . import open_loop
            self.S = open_loop.OpenLoopStack(self, _solver_factory)
            self.generated_header, self.model_hash = self.S.inner.generated_header, self.S.inner.model_hash
        else:
            self.generated_header = self._lower()
            self.model_hash = self.generated_header.rsplit('DOMPC_MODEL_HASH "', 1)[1].split('"')[0]
            factory = _solver_factory or HipIpmSolver
            rows = getattr(self, "_nlp_rows", None)
            ps_solver = self.structure
            if rows is not None:
                # node-local rows appended to nlp_cons: the solver's layout has extra row slots on every edge (nlp_route.ConstraintExtras)
                ps_solver = build_structure(**dict(self._structure_args, ne=self._structure_args["ne"] + rows.n_slots))
            self.S = factory(ps_solver, self.generated_header, self.model_hash, nlpsol_opts=self.settings.nlpsol_opts,
                             device=self.settings.gpu_index, max_batch=self.settings.max_batch,
                             block_threads=self.settings.block_threads)
            if rows is not None:
                from .solver import RowMappedSolver
                self.S = RowMappedSolver(self.S, rows.row_map(self.structure, ps_solver))
                self.n_opt_lagr = self._nlp_cons_lb.size          # (optimizer.py:1090: the number of rows of the concatenated nlp_cons)
        meta = {k: v for k, v in asdict(self.settings).items()}
        meta["structure_scenario"] = self.scenario_tree["structure_scenario"]
        self.data.set_meta(**meta)
        self.data._layouts = (self._opt_x_layout, self._opt_p_layout, self._opt_aux_layout)
        self.flags["setup"] = True

    # ------------------------------------------------------------------ runtime
    def set_initial_guess(self) -> None:
        assert self.flags["setup"] is True, "MPC was not setup yet. Please call MPC.setup()."
        self.opt_x_num["_x"] = self._x0.master / self._x_scaling.master
        self.opt_x_num["_u"] = self._u0.master / self._u_scaling.master
        self.opt_x_num["_z"] = self._z0.master / self._z_scaling.master
        self.flags["set_initial_guess"] = True

    def shard_tree(self, rank: int, world: int, cut_level=None, group=None, allreduce=None, native_rccl: bool = True) -> dict:
        """Shard the scenario tree of this controller's NLP over `world` ranks (one process per GPU):
        every rank builds the same MPC, calls shard_tree(rank, world) once after setup() and then make_step(x0) with
        identical x0; the ranks meet in torch.distributed all-reduces (backend nccl = RCCL) during the solve."""
        assert self.flags["setup"] is True, "MPC was not setup yet. Please call MPC.setup()."
        if self.rterm_expr is not None:
            raise NotImplementedError("structured HIP backend: tree sharding with a user-defined rterm expression")
        if self.structure.eps_global:
            raise NotImplementedError("structured HIP backend: tree sharding with nl_cons_single_slack")
        if self.structure.open_loop_stack:
            raise NotImplementedError("structured HIP backend: tree sharding with open_loop")
        if getattr(self.S, "row_mapped", False):
            raise NotImplementedError("structured HIP backend: tree sharding with rows appended to nlp_cons")
        if not getattr(self.S, "shard_capable", False):
            # the sharding-aware kernel is a second code object of the same model (build.py): swap the solver
            ctor = dict(self.S._ctor)
            self.S.close()
            self.S = HipIpmSolver(ctor.pop("structure"), ctor.pop("header_text"), ctor.pop("model_hash"), shard=True, **ctor)
        return self.S.enable_sharding(rank, world, cut_level=cut_level, group=group, allreduce=allreduce,
                                      native_rccl=native_rccl)

    def solve(self) -> None:
        """Optimizer.solve (optimizer.py:731-787) on the HIP solver."""
        assert self.flags["setup"] is True, "optimizer was not setup yet. Please call optimizer.setup()."
        r = self.S(x0=self.opt_x_num.master, lbx=self._lb_opt_x.master, ubx=self._ub_opt_x.master,
                   lbg=self._nlp_cons_lb, ubg=self._nlp_cons_ub, p=self.opt_p_num.master)
        self.opt_x_num.master[:] = r["x"]
        self.opt_x_num_unscaled.master[:] = r["x"] * self.opt_x_scaling.master
        self.opt_g_num = r["g"]
        self.lam_g_num = r["lam_g"]
        self.lam_x_num = r["lam_x"]
        self.solver_stats = self.S.stats()
        self.opt_aux_num.master[:] = self._eval_aux(self.opt_x_num_unscaled, self.opt_p_num)

    def _eval_aux(self, opt_x_unscaled: NumStruct, opt_p: NumStruct) -> np.ndarray:
        """opt_aux_expression_fun (_mpc.py:1277-1284, 1331): aux at every (k, s)."""
        m, ps, N = self.model, self.structure, self.settings.n_horizon
        X = opt_x_unscaled.master[:ps.off_z].reshape(N + 1, ps.S, ps.M + 1, ps.nx)[:N, :, -1, :]
        U = opt_x_unscaled.master[ps.off_u:ps.off_eps].reshape(N, ps.SU, ps.nu)
        TV = opt_p.master[ps.p_off_tvp:ps.p_off_p].reshape(N + 1, ps.ntvp)[:N]
        Pm = opt_p.master[ps.p_off_p:ps.p_off_uprev].reshape(ps.n_comb, ps.np_)
        cache = getattr(self, "_aux_index_cache", None)
        if cache is None:                     # index tables depend on the tree only: built once
            n_scen, n_br = ps.scenario_tree["n_scenarios"], ps.scenario_tree["n_branches"]
            boff = ps.scenario_tree["branch_offset"]
            src_s = np.zeros((N, ps.S), int)
            pidx = np.zeros((N, ps.S), int)
            for k in range(N):
                for s_ in range(ps.S):
                    s = min(s_, n_scen[k] - 1)
                    src_s[k, s_] = s
                    pidx[k, s_] = n_br[k] - 1 + boff[k][s]
            cache = self._aux_index_cache = (src_s, pidx)
        src_s, pidx = cache
        kk = np.repeat(np.arange(N), ps.S)
        Xf = X[kk, src_s.reshape(-1)].T
        Uf = U[kk, src_s.reshape(-1) if ps.SU == ps.S else 0].T        # (open_loop: `_u[k, 0]` for every scenario)
        Tf = TV[kk].T
        Pf = Pm[pidx.reshape(-1)].T
        if ps.nz:      # (_mpc.py:1277-1284: `_z[k, s, -1]`)
            Z = opt_x_unscaled.master[ps.off_z:ps.off_u].reshape(N, ps.S, max(ps.M, 1), ps.nz)[:, :, -1, :]
            Zf = Z[kk, src_s.reshape(-1)].T
        else:
            Zf = np.zeros((0, N * ps.S))
        out = m._aux_expression_fun.eval(Xf, Uf, Zf, Tf, Pf)[0]
        if out.ndim == 1:
            out = np.repeat(out[:, None], N * ps.S, axis=1)
        return out.T.reshape(-1)

    def make_step(self, x0) -> np.ndarray:
        assert self.flags["setup"] is True, "MPC was not setup yet. Please call MPC.setup()."
        if isinstance(x0, NumStruct):
            x0 = x0.master
        elif isinstance(x0, sym.DM):
            x0 = x0.arr
        elif not isinstance(x0, np.ndarray):
            raise Exception("Invalid type {} for x0. Must be {}".format(type(x0), (np.ndarray, sym.DM, NumStruct)))
        n_val = int(np.prod(x0.shape))
        assert n_val == self.model.n_x, "Wrong input with shape {}. Expected vector with {} elements".format(n_val, self.model.n_x)
        x0 = np.asarray(x0, dtype=float).reshape(-1)
        if not self.flags["set_initial_guess"]:
            warnings.warn("Intial guess for the MPC was not set. The solver call is likely to fail.")
            self.flags["set_initial_guess"] = True
        u_prev = self._u0.master.copy()
        tvp0 = self.tvp_fun(float(self._t0[0]))
        p0 = self.p_fun(float(self._t0[0]))
        t0 = self._t0.copy()
        self.opt_p_num["_x0"] = x0
        self.opt_p_num["_u_prev"] = u_prev
        self.opt_p_num.master[self.structure.p_off_tvp:self.structure.p_off_p] = tvp0.master
        self.opt_p_num.master[self.structure.p_off_p:self.structure.p_off_uprev] = p0.master
        self.solve()
        ps = self.structure
        u0 = self.opt_x_num.master[ps.iu(0, 0):ps.iu(0, 0) + ps.nu] * self._u_scaling.master
        z0 = np.zeros(0)
        aux0 = self.opt_aux_num.master[:self.model.n_aux]
        d = self.data
        d.update(_x=x0, _u=u0, _z=z0, _tvp=tvp0.master[:ps.ntvp], _p=p0.master[:ps.np_], _time=t0, _aux=aux0)
        d.update(opt_p_num=self.opt_p_num)
        if self.settings.store_full_solution:
            d.update(_opt_x_num=self.opt_x_num_unscaled, _opt_aux_num=self.opt_aux_num)
        if self.settings.store_lagr_multiplier:
            d.update(_lam_g_num=self.lam_g_num)
        if len(self.settings.store_solver_stats) > 0:
            d.update(**{k: v for k, v in self.solver_stats.items() if k in self.settings.store_solver_stats})
        self._t0 = self._t0 + self.settings.t_step
        self._x0.master[:] = x0
        self._u0.master[:] = u0
        return u0.reshape(-1, 1).copy()

    # ------------------------------------------------------------------ batched hot path (extension)
    def make_step_batch(self, X0: np.ndarray, U_prev: Optional[np.ndarray] = None,
                        opt_x_init: Optional[np.ndarray] = None) -> dict:
        """B independent MPC problems (same model/settings, different x0) in one device call.
        This is what do_mpc.sampling fans out over processes in the reference
        (/root/reference/do_mpc/sampling/_sampler.py:198-228)."""
        assert self.flags["setup"] is True, "MPC was not setup yet. Please call MPC.setup()."
        ps = self.structure
        X0 = np.asarray(X0, dtype=float).reshape(-1, ps.nx)
        B = X0.shape[0]
        P = np.tile(self.opt_p_num.master, (B, 1))
        P[:, :ps.nx] = X0
        tvp0 = self.tvp_fun(float(self._t0[0]))
        p0 = self.p_fun(float(self._t0[0]))
        P[:, ps.p_off_tvp:ps.p_off_p] = tvp0.master
        P[:, ps.p_off_p:ps.p_off_uprev] = p0.master
        P[:, ps.p_off_uprev:] = 0.0 if U_prev is None else np.asarray(U_prev, float).reshape(B, ps.nu)
        if opt_x_init is None:
            init = NumStruct(self._opt_x_layout, 0.0)
            Xi = np.tile(init.master, (B, 1))
            xblk = Xi[:, :ps.off_z].reshape(B, -1, ps.nx)
            xblk[:] = (X0 / self._x_scaling.master)[:, None, :]
            ublk = Xi[:, ps.off_u:ps.off_eps].reshape(B, -1, ps.nu)
            ublk[:] = (P[:, ps.p_off_uprev:] / self._u_scaling.master)[:, None, :]
        else:
            Xi = np.asarray(opt_x_init, float).reshape(B, ps.n_opt_x)
        r = self.S.solve_batch(Xi, self._lb_opt_x.master, self._ub_opt_x.master, self._nlp_cons_lb, self._nlp_cons_ub, P)
        r["u0"] = r["x"][:, ps.iu(0, 0):ps.iu(0, 0) + ps.nu] * self._u_scaling.master
        return r
