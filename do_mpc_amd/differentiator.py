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

Limits: the multiplier sensitivities cover the rows of g (`dlam_dp`), not the bound multipliers.
"""
from dataclasses import dataclass

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
class DifferentiatorSettings:
    """Names of the reference's settings are accepted (NLPDifferentiatorSettings, _nlpdifferentiator.py:40-116); the checks that
    need the dense reduced KKT matrix are not available here."""
    lin_solver: str = "hip"            # ignored: the solve is the controller's structured factorisation
    check_LICQ: bool = False
    check_SC: bool = False
    check_rank: bool = False
    track_residuals: bool = False
    lstsq_fallback: bool = False
    active_set_tol: float = 1e-6
    set_lam_zero: bool = False
    fd_step: float = 1e-6              # relative step of the central differences for parameters that enter nonlinearly


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
        self.optimizer = optimizer
        self.settings = DifferentiatorSettings(**kwargs)
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
        self.status = {}

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

    def differentiate(self):
        """Sensitivities at the solution stored in the controller (call after `make_step`).  Returns (dx_dp, dlam_dp);
        `sens_num["dxdp", indexf[...], indexf[...]]` afterwards, rows in UNSCALED variables like the reference."""
        mpc = self.optimizer
        if not mpc.solver_stats or not mpc.solver_stats.get("success", False):
            raise RuntimeError("DoMPCDifferentiator.differentiate(): no converged solution in the controller")
        x, lam, zl, zu, lb, ub, mu = self._point()
        lbg, ubg = mpc._nlp_cons_lb, mpc._nlp_cons_ub
        p0 = mpc.opt_p_num.master.copy()

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
        self.status = {"n_newton_solves": 1 + int(self._linear.sum()) + 2 * int((~self._linear).sum()),
                       "residual_step": float(np.max(np.abs(d0x)))}
        dxdp *= self.x_scaling_factors[:, None]                    # _nlpdifferentiator.py:851-853
        self.sens_num["dxdp"] = dxdp
        return dxdp.view(_Sens), dldp.view(_Sens)
