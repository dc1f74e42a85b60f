"""Parametric sensitivities of the MPC solution: d opt_x* / d opt_p from the structured KKT factorisation on the GPU.

Reference surface: `do_mpc.differentiator.DoMPCDifferentiator` (/root/reference/do_mpc/differentiator/
_nlpdifferentiator.py:730-870: `differentiate()` after a solve, `sens_num["dxdp", indexf[...], indexf[...]]`, rows rescaled
with `opt_x_scaling`, line 851-857).  The reference builds the dense symbolic KKT matrix of the reduced NLP (inactive
constraints removed, lines 287-301 and 469-555) and solves it with scipy / CasADi on the CPU.

Here the linear algebra is the controller's own: the primal-dual system of the interior-point method at the solution,
factorised by the per-edge condensing + tree Riccati kernels.  With F(v; p) = 0 the primal-dual
optimality conditions of the barrier problem at the final barrier parameter and K = dF/dv,

    dv/dp_j = -K^-1 dF/dp_j ,

and the Newton direction the kernels return at a point v is d(p) = -K(p)^-1 F(v; p), so

    dv/dp_j = [d(p + h e_j) - d(p)] / h        exactly for every parameter that enters F linearly (x0, u_prev: any h),
            ~ [d(p + h e_j) - d(p - h e_j)] / 2h  for the others (_p, _tvp: O(h^2), the matrix changes by O(h |F|) ~ 0).

All columns (+1 residual direction) in one batched `dompc_newton_steps_at_solution` call, all at the solution point - no symbolic KKT matrix, no dense
solve.  Models with nl_cons rows: the slack variables of the rows take their values at the solution (s = d(x), multipliers
mu / distance) inside the call, the soft-constraint variables `_eps` are ordinary decision variables.
Strict complementarity is assumed like in the reference (`check_SC`): a bound whose multiplier is not clearly separated
from zero gets the sensitivities of the barrier problem (a smoothed active set), not a kink.

Round 5: the reference's treatment of the active set (`_get_active_constraints`, `_extract_active_primal_dual_solution`,
_nlpdifferentiator.py:347-428) restated on top of that: constraints and bounds are classified by their distance to the bound
(`settings.active_set_tol`), `status.SC` / `status.LICQ` are the reference's checks (`_check_SC`, `_check_LICQ`, lines 620-649; the
constraint Jacobian of the LICQ check comes from batched model-evaluation sweeps on the GPU, `constraint_jacobian`), and with
`settings.active_set_reduction` the inactive bounds leave the system (multiplier exactly zero, as the reference removes their rows)
while the active ones are held like equalities - the sensitivities of the reduced NLP instead of those of the barrier problem.
`NLPDifferentiator` (bottom of the file) is the reference's stand-alone class for an arbitrary small NLP given as symbolic expressions:
dense KKT matrix on the host, like the reference's (no structure to exploit there).

Limits: the multiplier sensitivities cover the rows of g (`dlam_dp`), not the bound multipliers.
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np


class indexf_type:
    """`casadi.tools.indexf[...]`: a power index handed on to the structure it is applied to."""

    def __getitem__(self, key):
        return _IndexF(key if isinstance(key, tuple) else (key,))


class _IndexF:
    def __init__(self, key):
        self.key = key


indexf = indexf_type()


class _Sens(np.ndarray):
    def full(self):
        return np.asarray(self)


@dataclass
class NLPDifferentiatorSettings:
    """The reference's settings (NLPDifferentiatorSettings, differentiator/helper.py:13-69), same names and defaults except
    `lin_solver` (the solve is the controller's structured factorisation on the GPU) and `check_LICQ` (off by default: it needs the
    whole constraint Jacobian - the reference's own warning: "computationally demanding ... debugging purposes")."""
    lin_solver: str = "hip"            # ignored by DoMPCDifferentiator; NLPDifferentiator: 'scipy' (dense LU) or 'lstsq'
    check_LICQ: bool = False
    check_SC: bool = True
    track_residuals: bool = True
    check_rank: bool = False
    lstsq_fallback: bool = False
    active_set_tol: float = 1e-6
    set_lam_zero: bool = False
    # extensions
    active_set_reduction: bool = False  # DoMPCDifferentiator: sensitivities of the active-set-reduced NLP (see module docstring)
    fd_step: float = 1e-6              # relative step of the central differences for parameters that enter nonlinearly


DifferentiatorSettings = NLPDifferentiatorSettings          # (name used until round 4)


@dataclass
class NLPDifferentiatorStatus:
    """differentiator/helper.py:72-117"""
    LICQ: Optional[bool] = None
    SC: Optional[bool] = None
    residuals: Optional[float] = None
    lse_solved: bool = False
    full_rank: Optional[bool] = None
    sym_KKT: bool = False
    reduced_nlp: bool = False
    # extensions: what the last differentiate() saw
    n_active_g: Optional[int] = None
    n_active_x: Optional[int] = None
    n_newton_solves: Optional[int] = None
    residual_step: Optional[float] = None

    def __getitem__(self, key):          # (round 2-4 code read the status as a dict)
        return getattr(self, key)


def active_constraints(x, g, lbx, ubx, lbg, ubg, tol):
    """_get_active_constraints (_nlpdifferentiator.py:347-394): by the distance of the PRIMAL solution to the bounds"""
    x, g = np.ravel(x), np.ravel(g)
    g_act = (np.abs(g - np.ravel(lbg)) <= tol) | (np.abs(g - np.ravel(ubg)) <= tol)
    x_act = (np.abs(x - np.ravel(lbx)) <= tol) | (np.abs(x - np.ravel(ubx)) <= tol)
    return np.where(~g_act)[0], np.where(~x_act)[0], np.where(g_act)[0], np.where(x_act)[0]


def check_sc(lam, where_cons_active, tol) -> bool:
    """_check_SC (_nlpdifferentiator.py:636-649): every multiplier of the active set is at least `tol` in magnitude"""
    return bool(np.all(np.abs(np.ravel(lam)[where_cons_active]) >= tol))


def rows_independent(M, tol=1e-10) -> bool:
    """Do the rows of the sparse / dense matrix M have full rank?  (np.linalg.matrix_rank of the reference, _check_LICQ lines
    620-634, for small systems; larger ones: sparse LU of M M' - rank deficient exactly when a pivot collapses.)"""
    import scipy.sparse as sps
    n = M.shape[0]
    if n == 0:
        return True
    if n > M.shape[1]:
        return False
    if n <= 1500:
        D = M.toarray() if sps.issparse(M) else np.asarray(M)
        return int(np.linalg.matrix_rank(D)) == n
    import scipy.sparse.linalg as spla
    M = sps.csr_matrix(M)
    rn = np.sqrt(np.asarray(M.multiply(M).sum(axis=1)).ravel())
    if np.any(rn == 0.0):
        return False
    M = sps.diags(1.0 / rn) @ M
    try:
        lu = spla.splu(sps.csc_matrix(M @ M.T), diag_pivot_thresh=1.0)
    except RuntimeError:
        return False
    d = np.abs(lu.U.diagonal())
    return bool(d.min() > tol * d.max())


class _SensNum:
    def __init__(self, owner):
        self._o = owner
        self.dxdp = None

    def __setitem__(self, key, value):
        if key != "dxdp":
            raise KeyError(key)
        self.dxdp = np.asarray(value, float)

    def __getitem__(self, key):
        if key == "dxdp":
            return self.dxdp.view(_Sens)
        name, ix, ip = key
        if name != "dxdp" or self.dxdp is None:
            raise KeyError(key)
        rows = self._o.optimizer._opt_x_layout.resolve(ix.key).ravel()
        cols = self._o.optimizer._opt_p_layout.resolve(ip.key).ravel()
        return self.dxdp[np.ix_(rows, cols)].view(_Sens)


class DoMPCDifferentiator:
    def __init__(self, optimizer, **kwargs):
        if getattr(getattr(optimizer, "S", None), "row_mapped", False):
            raise NotImplementedError("structured HIP backend: DoMPCDifferentiator on an NLP with rows appended to nlp_cons (the solver's "
                                      "own KKT system has the internal row layout, solver.RowMappedSolver)")
        self.optimizer = optimizer
        self.settings = NLPDifferentiatorSettings(**kwargs)
        ps = optimizer.structure
        if getattr(ps, "eps_global", False):
            raise NotImplementedError("structured HIP backend: DoMPCDifferentiator with nl_cons_single_slack (the Newton steps at the "
                                      "solution do not carry the Schur complement of the shared slack variables)")
        if getattr(ps, "open_loop_stack", False):
            raise NotImplementedError("structured HIP backend: DoMPCDifferentiator with open_loop and several scenarios")
        self.x_scaling_factors = optimizer.opt_x_scaling.master.copy()
        self.sens_num = _SensNum(self)
        self.n_x, self.n_p, self.n_g = ps.n_opt_x, ps.n_opt_p, ps.n_g
        # parameters that enter the optimality conditions linearly: x0 (initial-condition rows) and u_prev (rterm gradient)
        lay = optimizer._opt_p_layout
        lin = np.zeros(self.n_p, bool)
        lin[lay.resolve(("_x0",)).ravel()] = True
        lin[lay.resolve(("_u_prev",)).ravel()] = True
        self._linear = lin
        self.status = NLPDifferentiatorStatus(sym_KKT=True)      # (nothing symbolic to prepare: the KKT system is the solver's)
        self._jac_solver = None

    # ------------------------------------------------------------------------------------------------
    def _point(self):
        """The solution as the primal-dual point of the barrier problem in the solver's own (objective-unscaled) variables."""
        mpc = self.optimizer
        st = mpc.solver_stats
        x = mpc.opt_x_num.master.copy()
        lam = np.asarray(mpc.lam_g_num, float).copy()
        mu = float(st["mu"]) / float(st.get("obj_scaling", 1.0))
        lb, ub = mpc._lb_opt_x.master.copy(), mpc._ub_opt_x.master.copy()
        hl, hu = np.isfinite(lb), np.isfinite(ub)
        opts = getattr(mpc.S, "options", None)                       # the relaxation the solver applied to the bounds
        relax = float(getattr(opts, "bound_relax_factor", 1e-8))
        cvt = float(getattr(opts, "constr_viol_tol", 1e-4))
        lb[hl] -= np.minimum(cvt, relax * np.maximum(1.0, np.abs(lb[hl])))
        ub[hu] += np.minimum(cvt, relax * np.maximum(1.0, np.abs(ub[hu])))
        dl, du = np.where(hl, x - lb, 1.0), np.where(hu, ub - x, 1.0)
        if dl.min() <= 0 or du.min() <= 0:
            raise RuntimeError("DoMPCDifferentiator: the stored solution is not strictly inside its (relaxed) bounds")
        zl, zu = np.where(hl, mu / dl, 0.0), np.where(hu, mu / du, 0.0)
        return x, lam, zl, zu, lb, ub, mu

    # ------------------------------------------------------------------------------------------------
    def constraint_jacobian(self, x=None, lam=None, p=None, chunk: int = 512):
        """dg/dx (scipy.sparse CSR, n_g x n_opt_x, scaled variables) at a point, column by column from central differences of BATCHED
        model-evaluation sweeps on the GPU (`dompc_sweep_batch_device`: one launch evaluates g for hundreds of perturbed iterates; a
        row that does not depend on a variable returns bit-identical values, so the sparsity pattern is exact).  Used by the LICQ
        check; the solve never needs it."""
        import scipy.sparse as sps
        mpc = self.optimizer
        x = mpc.opt_x_num.master.copy() if x is None else np.asarray(x, float)
        lam = np.asarray(mpc.lam_g_num, float).copy() if lam is None else np.asarray(lam, float)
        p = mpc.opt_p_num.master.copy() if p is None else np.asarray(p, float)
        S = self._sweep_solver(chunk)
        n, m = self.n_x, self.n_g
        h = 1e-6 * np.maximum(1.0, np.abs(x))
        rows, cols, vals = [], [], []
        emu = S._host_emulation
        if not emu:
            import torch
            dev = torch.device("cuda", mpc.settings.gpu_index)
        for j0 in range(0, n, chunk):
            js = np.arange(j0, min(n, j0 + chunk))
            X = np.repeat(x[None, :], 2 * js.size, axis=0)
            X[np.arange(js.size), js] += h[js]
            X[js.size + np.arange(js.size), js] -= h[js]
            L = np.repeat(lam[None, :], 2 * js.size, axis=0)
            P = np.repeat(p[None, :], 2 * js.size, axis=0)
            if emu:
                G = np.empty((2 * js.size, m))
                S.sweep_batch_device(2 * js.size, X.ctypes.data, L.ctypes.data, P.ctypes.data, G.ctypes.data, 0)
            else:
                tX, tL, tP = (torch.from_numpy(a).to(dev) for a in (X, L, P))
                tG = torch.empty((2 * js.size, m), dtype=torch.float64, device=dev)
                torch.cuda.synchronize(dev)
                S.sweep_batch_device(2 * js.size, tX.data_ptr(), tL.data_ptr(), tP.data_ptr(), tG.data_ptr(), 0,
                                     stream=torch.cuda.current_stream(dev).cuda_stream)
                torch.cuda.synchronize(dev)
                G = tG.cpu().numpy()
            D = (G[:js.size] - G[js.size:]) / (2.0 * h[js])[:, None]
            c, r = np.nonzero(D)
            rows.append(r); cols.append(js[c]); vals.append(D[c, r])
        return sps.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m, n))

    def _sweep_solver(self, chunk):
        """a second handle of the same problem class with enough workspace slots for a batch of sweeps (the controller's own handle
        has `max_batch` slots - one for a plain MPC)"""
        S = self.optimizer.S
        if getattr(S, "_host_emulation", False) or S.num_slots >= 64:
            return S
        if self._jac_solver is None:
            ctor = dict(S._ctor)
            ctor["max_batch"] = max(int(ctor.get("max_batch", 1)), 2 * chunk)
            self._jac_solver = type(S)(ctor.pop("structure"), ctor.pop("header_text"), ctor.pop("model_hash"), **ctor)
        return self._jac_solver

    def _check_licq(self, x, lam, p, g_act, x_act) -> bool:
        """_check_LICQ (_nlpdifferentiator.py:620-634): the gradients of the active constraints [g; x] are linearly independent"""
        import scipy.sparse as sps
        J = self.constraint_jacobian(x, lam, p)
        E = sps.csr_matrix((np.ones(x_act.size), (np.arange(x_act.size), x_act)), shape=(x_act.size, self.n_x))
        return rows_independent(sps.vstack([J[g_act], E], format="csr"))

    def differentiate(self):
        """Sensitivities at the solution stored in the controller (call after `make_step`).  Returns (dx_dp, dlam_dp);
        `sens_num["dxdp", indexf[...], indexf[...]]` afterwards, rows in UNSCALED variables like the reference."""
        mpc = self.optimizer
        if not mpc.solver_stats or not mpc.solver_stats.get("success", False):
            raise RuntimeError("DoMPCDifferentiator.differentiate(): no converged solution in the controller")
        x, lam, zl, zu, lb, ub, mu = self._point()
        lbg, ubg = mpc._nlp_cons_lb, mpc._nlp_cons_ub
        p0 = mpc.opt_p_num.master.copy()
        st, cfg = self.status, self.settings
        st.lse_solved = False
        # ---- active set of the solution (_nlpdifferentiator.py:347-428): distances of the primal solution to the ORIGINAL bounds
        g_num = np.asarray(mpc.opt_g_num, float).reshape(-1)
        _, x_in, g_act, x_act = active_constraints(x, g_num, mpc._lb_opt_x.master, mpc._ub_opt_x.master, lbg, ubg, cfg.active_set_tol)
        where_cons_active = np.concatenate([g_act, x_act + self.n_g])
        st.n_active_g, st.n_active_x = int(g_act.size), int(x_act.size)
        self.where_cons_active = where_cons_active
        if cfg.check_SC:
            st.SC = check_sc(np.concatenate([lam, np.asarray(mpc.lam_x_num, float).reshape(-1)]), where_cons_active, cfg.active_set_tol)
        if cfg.check_LICQ:
            st.LICQ = self._check_licq(x, lam, p0, g_act, x_act)
        point0 = None
        if cfg.active_set_reduction:
            point0 = (zl, zu, lb, ub, lbg, ubg)          # the solver's own point: what status.residual_step is measured at (below)
            # inactive bounds leave the system (the reference removes their rows and columns; set_lam_zero: their multipliers are
            # exactly zero), active ones are held like equalities: Sigma = z / distance >= 1e9 (z >= 1e3, distance <= tol: the size a converged active bound has)
            zl, zu, lb, ub = zl.copy(), zu.copy(), lb.copy(), ub.copy()
            zl[x_in], zu[x_in], lb[x_in], ub[x_in] = 0.0, 0.0, -np.inf, np.inf
            lo = np.abs(x - mpc._lb_opt_x.master) <= cfg.active_set_tol
            up = np.abs(x - mpc._ub_opt_x.master) <= cfg.active_set_tol
            zl[lo] = np.maximum(zl[lo], 1e3)
            zu[up] = np.maximum(zu[up], 1e3)
            # inequality rows of g (nl_cons): an inactive row loses its bounds (its slack is then a free variable: the row leaves
            # the system), an active one becomes an equality at its value
            lbg, ubg = np.array(lbg, float), np.array(ubg, float)
            ineq = lbg != ubg
            act = np.zeros(self.n_g, bool)
            act[g_act] = True
            lbg[ineq & ~act], ubg[ineq & ~act] = -np.inf, np.inf
            lbg[ineq & act] = ubg[ineq & act] = g_num[ineq & act]
            st.reduced_nlp = True

        # every direction in ONE batched call (dompc_newton_steps_at_solution: one workgroup per parameter vector): the residual
        # direction at p0, one forward step per parameter that enters linearly (exact), a central pair for the others
        rows, plan = [p0.copy()], []
        for j in range(self.n_p):
            if self._linear[j]:
                h = max(1.0, abs(p0[j]))
                p = p0.copy(); p[j] = p0[j] + h
                plan.append((j, h, len(rows), -1)); rows.append(p)
            else:
                h = self.settings.fd_step * max(1.0, abs(p0[j]))
                pp, pm = p0.copy(), p0.copy()
                pp[j], pm[j] = p0[j] + h, p0[j] - h
                plan.append((j, h, len(rows), len(rows) + 1)); rows.extend([pp, pm])
        DX, DL = mpc.S.newton_steps_at_solution(x, lam, zl, zu, lb, ub, lbg, ubg, np.array(rows), mu)
        if not np.all(np.isfinite(DX)):
            raise RuntimeError("DoMPCDifferentiator: the KKT matrix at the solution has the wrong inertia")
        d0x, d0l = DX[0], DL[0]
        dxdp = np.zeros((self.n_x, self.n_p))
        dldp = np.zeros((self.n_g, self.n_p))
        for j, h, ip, im in plan:
            if im < 0:
                dxdp[:, j], dldp[:, j] = (DX[ip] - d0x) / h, (DL[ip] - d0l) / h
            else:
                dxdp[:, j], dldp[:, j] = (DX[ip] - DX[im]) / (2 * h), (DL[ip] - DL[im]) / (2 * h)
        st.n_newton_solves = 1 + int(self._linear.sum()) + 2 * int((~self._linear).sum())
        used = np.ones(self.n_x, bool)
        used[np.asarray(mpc.structure.tables["dummy_idx"], dtype=int)] = False      # (variables that appear nowhere in the NLP)
        if point0 is not None:
            # (with held bounds - multipliers raised to 1e3, active rows pinned - the point handed to the solves above is no KKT
            #  point of the barrier problem any more, its Newton direction d0 does not measure the quality of the solution: one more solve at
            #  the UNMODIFIED point for the status; the sensitivities are differences of directions and do not change)
            d0x = mpc.S.newton_steps_at_solution(x, lam, *point0, p0[None, :].copy(), mu)[0][0]
            st.n_newton_solves += 1
        st.residual_step = float(np.max(np.abs(d0x[used])))
        st.lse_solved = True
        st.full_rank = True if cfg.check_rank else st.full_rank      # (every structured solve above succeeded with delta_w = 0: the inertia is exact)
        if cfg.track_residuals:
            # the reference reports |A S + B| of its dense solve (_track_residuals); the structured solve has no such matrix in memory -
            # what is reported is the Newton direction at the solution itself, i.e. the residual of the optimality conditions the
            # sensitivities are linearised about, in units of the step (0 at an exact solution)
            st.residuals = st.residual_step
        dxdp *= self.x_scaling_factors[:, None]                    # _nlpdifferentiator.py:851-853
        self.sens_num["dxdp"] = dxdp
        return dxdp.view(_Sens), dldp.view(_Sens)


class NLPDifferentiator:
    """The reference's stand-alone differentiator for an arbitrary NLP (/root/reference/do_mpc/differentiator/
    _nlpdifferentiator.py:17-728) given as SYMBOLIC expressions of do_mpc_amd.sym: `nlp = {'x', 'p', 'f', 'g'}`,
    `nlp_bounds = {'lbx', 'ubx', 'lbg', 'ubg'}`; `differentiate(nlp_sol, p_num)` with `nlp_sol = {'x', 'g', 'lam_g', 'lam_x'}` (multiplier
    signs: L = f + lam_g' g + lam_x' x, line 287) returns (dx_dp, dlam_dp).  Same steps as the reference: unused variables and parameters
    are removed (`_remove_unused_sym_vars`), A = hess_z L and B = d grad_z L / dp are formed symbolically once (`_prepare_sensitivity_
    matrices`), per call the active set is read off the primal solution, rows and columns of inactive constraints are dropped
    (`_reduce_sensitivity_matrices`) and the dense system A S = -B is solved.  Host code, like the reference's - an arbitrary small NLP
    has no stage structure the GPU solver could use; the controller's NLPs go through DoMPCDifferentiator."""

    def __init__(self, nlp: dict, nlp_bounds: dict, **kwargs):
        from . import sym
        if not isinstance(nlp, dict):
            raise ValueError("nlp must be a dictionary.")
        if not isinstance(nlp_bounds, dict):
            raise ValueError("nlp_bounds must be a dictionary.")
        if not set(nlp.keys()).issuperset({"f", "x", "p", "g"}):
            raise ValueError("nlp must contain keys {}.".format(["f", "x", "p", "g"]))
        if not set(nlp_bounds.keys()).issuperset({"lbx", "ubx", "lbg", "ubg"}):
            raise ValueError("nlp_bounds must contain keys {}.".format(["lbx", "ubx", "lbg", "ubg"]))
        self._sym = sym
        self.nlp = {k: sym._sx(v) for k, v in nlp.items()}
        self.nlp_bounds = {k: np.asarray(getattr(v, "arr", v), float).reshape(-1) for k, v in nlp_bounds.items()}
        self.status = NLPDifferentiatorStatus()
        kwargs.setdefault("lin_solver", "scipy")
        kwargs.setdefault("check_LICQ", True)
        self.settings = NLPDifferentiatorSettings(**kwargs)
        self._prepare_differentiator()

    # ---- preparation (once)
    def _prepare_differentiator(self):
        sym = self._sym
        fg = list(self.nlp["f"].nodes()) + list(self.nlp["g"].nodes())
        free = {id(n) for n in sym.free_symbols(fg)}
        det_x = np.array([i for i, n in enumerate(self.nlp["x"].nodes()) if id(n) in free], dtype=int)
        det_p = np.array([i for i, n in enumerate(self.nlp["p"].nodes()) if id(n) in free], dtype=int)
        self.n_x_unreduced, self.n_p_unreduced = self.nlp["x"].numel(), self.nlp["p"].numel()
        self.det_sym_idx_dict = {"opt_x": det_x, "opt_p": det_p}
        self.status.reduced_nlp = det_x.size < self.n_x_unreduced or det_p.size < self.n_p_unreduced
        xs = sym.SX([self.nlp["x"].nodes()[i] for i in det_x], (det_x.size, 1))
        ps = sym.SX([self.nlp["p"].nodes()[i] for i in det_p], (det_p.size, 1))
        self._x, self._p = xs, ps
        self._lbx, self._ubx = self.nlp_bounds["lbx"][det_x], self.nlp_bounds["ubx"][det_x]
        self.n_x, self.n_p, self.n_g = det_x.size, det_p.size, self.nlp["g"].numel()
        lam_g, lam_x = sym.SX.sym("lam_g", self.n_g), sym.SX.sym("lam_x", self.n_x)
        z = sym.vertcat(xs, lam_g, lam_x)
        L = self.nlp["f"] + sym.dot(lam_g, self.nlp["g"]) + sym.dot(lam_x, xs)                      # line 287
        A, gz = sym.hessian(L, z)
        self.A_func = sym.Function("A", [z, ps], [A])
        self.B_func = sym.Function("B", [z, ps], [sym.jacobian(gz, ps)])
        self.cons_grad_func = sym.Function("cons_grad", [xs, ps], [sym.jacobian(sym.vertcat(self.nlp["g"], xs), xs)])
        self.status.sym_KKT = True

    # ---- per solution
    def _solve_linear_system(self, A, B, lin_solver):
        self.status.lse_solved = False
        try:
            if lin_solver == "lstsq":
                S = np.linalg.lstsq(A, -B, rcond=None)[0]
            else:
                S = np.linalg.solve(A, -B)
            if not np.all(np.isfinite(S)):
                raise np.linalg.LinAlgError("non-finite solution")
            self.status.lse_solved = True
        except np.linalg.LinAlgError:
            S = np.full((A.shape[0], B.shape[1]), np.nan)       # (the reference returns NaNs when the system cannot be solved)
        return S

    def differentiate(self, nlp_sol: dict, p_num):
        if not isinstance(nlp_sol, dict):
            raise ValueError("nlp_sol must be a dictionary.")
        if not set(nlp_sol.keys()).issuperset({"x", "lam_g", "lam_x", "g"}):
            raise ValueError("nlp_sol must contain keys {}.".format(["x", "lam_g", "lam_x", "g"]))
        vec = lambda v: np.asarray(getattr(v, "arr", v), float).reshape(-1)      # noqa: E731
        dx_, dp_ = self.det_sym_idx_dict["opt_x"], self.det_sym_idx_dict["opt_p"]
        p_full = vec(p_num)
        if p_full.size != self.n_p_unreduced:
            raise ValueError("p_num must have length {}.".format(self.n_p_unreduced))
        x, lam_x, p = vec(nlp_sol["x"])[dx_], vec(nlp_sol["lam_x"])[dx_], p_full[dp_]
        g, lam_g = vec(nlp_sol["g"]), vec(nlp_sol["lam_g"])
        cfg, st = self.settings, self.status
        g_in, x_in, g_act, x_act = active_constraints(x, g, self._lbx, self._ubx, self.nlp_bounds["lbg"], self.nlp_bounds["ubg"],
                                                      cfg.active_set_tol)
        where_act = np.concatenate([g_act, x_act + self.n_g])
        lam = np.concatenate([lam_g, lam_x])
        if cfg.set_lam_zero:
            lam[np.concatenate([g_in, x_in + self.n_g])] = 0.0
        z = np.concatenate([x, lam])
        if cfg.check_LICQ:
            cg = self.cons_grad_func.eval(x, p)[0].reshape(self.n_g + self.n_x, self.n_x, order="F")[where_act]
            st.LICQ = rows_independent(cg)
        if cfg.check_SC:
            st.SC = check_sc(lam, where_act, cfg.active_set_tol)
        nz = self.n_x + self.n_g + self.n_x
        A = self.A_func.eval(z, p)[0].reshape(nz, nz, order="F")
        B = self.B_func.eval(z, p)[0].reshape(nz, self.n_p, order="F")
        keep = np.concatenate([np.arange(self.n_x), where_act + self.n_x])
        A, B = A[np.ix_(keep, keep)], B[keep]
        if cfg.check_rank:
            st.full_rank = bool(np.linalg.matrix_rank(A) == A.shape[0])
        S = self._solve_linear_system(A, B, cfg.lin_solver)
        if not st.lse_solved and cfg.lstsq_fallback:
            S = self._solve_linear_system(A, B, "lstsq")
        if cfg.track_residuals:
            st.residuals = float(np.linalg.norm(A @ S + B)) if st.lse_solved else None
        dx_red = S[:self.n_x]
        dlam_red = np.zeros((self.n_g + self.n_x, self.n_p))
        dlam_red[where_act] = S[self.n_x:]
        # back to the full variables / parameters (_map_param_sens_to_full)
        dx_dp = np.zeros((self.n_x_unreduced, self.n_p_unreduced))
        dx_dp[np.ix_(dx_, dp_)] = dx_red
        dlam_dp = np.zeros((self.n_g + self.n_x_unreduced, self.n_p_unreduced))
        rows = np.concatenate([np.arange(self.n_g), dx_ + self.n_g])
        dlam_dp[np.ix_(rows, dp_)] = dlam_red
        return dx_dp.view(_Sens), dlam_dp.view(_Sens)
