"""Checks of do_mpc_amd.differentiator shared by the host-emulation (CPU CI) and the HIP (-m gpu) test modules.

SURVEY.md 8(f) row 3.  The reference has no golden vectors for its differentiator (testing/ holds none), so the sensitivities
are pinned two ways: (1) against the oracle's general sparse LU of the same primal-dual system with the parameter
derivative formed from the oracle's own NLP functions, (2) against central finite differences of complete re-solves."""
import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla

import parity_common as pc
from do_mpc_amd.differentiator import DoMPCDifferentiator, indexf
from do_mpc_amd.examples import CASES


def solved(make_mpc, name, **over):
    ex = CASES[name]
    mpc = make_mpc(name, **over)
    mpc.x0 = ex.X0
    mpc.set_initial_guess()
    mpc.make_step(ex.X0)
    assert mpc.solver_stats["success"]
    return mpc


def oracle_sensitivity(mpc, name, cols, **over):
    """d opt_x / d opt_p[cols] (scaled variables) from a sparse LU of the oracle's KKT matrix at the product's solution,
    for parameters that enter linearly (difference of two oracle Newton directions, exact)."""
    nlp = pc.oracle_nlp(name, **over)
    nd = DoMPCDifferentiator(mpc)
    x, lam, zl, zu, lb, ub, mu = nd._point()
    p0 = mpc.opt_p_num.master.copy()
    hl, hu = np.isfinite(lb), np.isfinite(ub)
    dl, du = np.where(hl, x - lb, 1.0), np.where(hu, ub - x, 1.0)
    sig = zl / dl * hl + zu / du * hu
    dummy = np.asarray(mpc.structure.tables["dummy_idx"])
    pin = np.zeros(x.size)
    pin[dummy] = (sig[dummy] == 0)
    W, A = nlp.hess(x, p0, 1.0, lam), nlp.jac(x, p0)
    K = sps.bmat([[W + sps.diags(sig + pin), A.T], [A, None]], format="csc")
    lu = spla.splu(K)

    def direction(p):
        rx = nlp.grad(x, p) + nlp.jac(x, p).T @ lam - np.where(hl, mu / dl, 0.0) + np.where(hu, mu / du, 0.0)
        rhs = -np.concatenate([rx, nlp.g(x, p) - nlp.lbg])
        sol = lu.solve(rhs)
        for _ in range(2):
            sol += lu.solve(rhs - K @ sol)
        return sol[:x.size]

    d0 = direction(p0)
    out = np.zeros((x.size, len(cols)))
    for k, j in enumerate(cols):
        p = p0.copy()
        h = max(1.0, abs(p0[j]))
        p[j] += h
        out[:, k] = (direction(p) - d0) / h
    return out


def check_against_oracle_kkt(make_mpc, name, **over):
    mpc = solved(make_mpc, name, **over)
    nd = DoMPCDifferentiator(mpc)
    dxdp, dldp = nd.differentiate()
    lay = mpc._opt_p_layout
    cols = np.concatenate([lay.resolve(("_x0",)).ravel(), lay.resolve(("_u_prev",)).ravel()])
    ref = oracle_sensitivity(mpc, name, cols, **over) * mpc.opt_x_scaling.master[:, None]
    used = np.ones(mpc.structure.n_opt_x, bool)
    used[mpc.structure.tables["dummy_idx"]] = False
    err = np.max(np.abs(np.asarray(dxdp)[used][:, cols] - ref[used]))
    assert err < 1e-6 * max(1.0, np.max(np.abs(ref[used]))), err
    assert dldp.shape == (mpc.structure.n_g, mpc.structure.n_opt_p)
    return mpc, nd


def check_against_resolves(make_mpc, name, p_keys, rtol=2e-3, **over):
    """du0/dp against central differences of complete cold re-solves, one opt_p entry per key."""
    ex = CASES[name]
    mpc = solved(make_mpc, name, **over)
    nd = DoMPCDifferentiator(mpc)
    nd.differentiate()
    lay = mpc._opt_p_layout
    p0 = mpc.opt_p_num.master.copy()
    for key in p_keys:
        j = int(lay.resolve(key).ravel()[0])
        du0 = np.asarray(nd.sens_num["dxdp", indexf["_u", 0, 0], indexf[key]])[:, 0]
        h = 1e-5 * max(1.0, abs(p0[j]))
        us = []
        for sgn in (1.0, -1.0):
            m2 = make_mpc(name, **over)
            m2.x0 = ex.X0
            m2.set_initial_guess()
            m2.opt_p_num.master[:] = p0
            m2.opt_p_num.master[j] = p0[j] + sgn * h
            m2.solve()
            assert m2.solver_stats["success"]
            us.append(m2.opt_x_num_unscaled["_u", 0, 0].copy().ravel() if hasattr(m2.opt_x_num_unscaled["_u", 0, 0], "copy")
                      else np.asarray(m2.opt_x_num_unscaled["_u", 0, 0]).ravel())
        fd = (us[0] - us[1]) / (2 * h)
        scale = max(np.max(np.abs(fd)), 1e-8)
        assert np.max(np.abs(fd - du0)) < rtol * scale + 1e-7, (key, fd, du0)


def check_batched_directions_equal_single_rows(make_mpc, name, max_batch=8, **over):
    """`dompc_newton_steps_at_solution` with SEVERAL workspace slots (max_batch > 1: one workgroup and one slot per parameter
    vector) against the single-row entry point `dompc_newton_step_at_solution`, row by row (the batched call used to
    factorise slot 0 with parameter row 0 in every workgroup; with max_batch = 1 the batch degenerates to a loop and hides it)."""
    mpc = solved(make_mpc, name, max_batch=max_batch, **over)
    assert mpc.S.num_slots > 1 or mpc.S._host_emulation      # (the host emulation runs the rows one after the other in one slot)
    nd = DoMPCDifferentiator(mpc)
    x, lam, zl, zu, lb, ub, mu = nd._point()
    lbg, ubg = mpc._nlp_cons_lb, mpc._nlp_cons_ub
    p0 = mpc.opt_p_num.master.copy()
    lay = mpc._opt_p_layout
    cols = np.concatenate([lay.resolve(("_x0",)).ravel(), lay.resolve(("_u_prev",)).ravel()])
    rows = [p0.copy()]
    for j in cols:
        p = p0.copy()
        p[j] += max(1.0, abs(p0[j]))
        rows.append(p)
    rows = np.array(rows)
    assert len(rows) > 2
    DX, DL = mpc.S.newton_steps_at_solution(x, lam, zl, zu, lb, ub, lbg, ubg, rows, mu)
    differ = 0.0
    for i, p in enumerate(rows):
        dx, dl = mpc.S.newton_step_at_solution(x, lam, zl, zu, lb, ub, lbg, ubg, p, mu)
        sc = max(1.0, np.max(np.abs(dx)))
        assert np.max(np.abs(DX[i] - dx)) <= 1e-12 * sc, (i, np.max(np.abs(DX[i] - dx)))
        assert np.max(np.abs(DL[i] - dl)) <= 1e-12 * max(1.0, np.max(np.abs(dl))), i
        if i:
            differ = max(differ, np.max(np.abs(DX[i] - DX[0])))
    assert differ > 1e-6            # (the rows are not all the direction of parameter row 0)
    # ... and the sensitivities of the batched differentiator equal those of a controller with a single slot
    dxdp, _ = nd.differentiate()
    mpc1 = solved(make_mpc, name, **over)
    dxdp1, _ = DoMPCDifferentiator(mpc1).differentiate()
    used = np.ones(mpc.structure.n_opt_x, bool)
    used[mpc.structure.tables["dummy_idx"]] = False
    ref = np.asarray(dxdp1)[used][:, cols]
    assert np.max(np.abs(np.asarray(dxdp)[used][:, cols] - ref)) <= 1e-7 * max(1.0, np.max(np.abs(ref)))


def check_status_and_jacobian(make_mpc, name, **over):
    """Round 5: the reference's status object (differentiator/helper.py:72-117) and its checks.  The constraint Jacobian the LICQ check
    uses (batched sweeps of the kernels + central differences) against the oracle's analytic one; LICQ / SC flags of a regular solution."""
    from do_mpc_amd.differentiator import NLPDifferentiatorStatus, active_constraints
    mpc = solved(make_mpc, name, **over)
    nd = DoMPCDifferentiator(mpc, check_LICQ=True, check_SC=True, check_rank=True)
    assert isinstance(nd.status, NLPDifferentiatorStatus) and nd.status.LICQ is None and nd.status.SC is None and not nd.status.lse_solved
    nd.differentiate()
    st = nd.status
    assert st.lse_solved and st.LICQ is True and st.full_rank is True and st.sym_KKT and st.residuals is not None and st.residuals < 1e-5
    assert isinstance(st.SC, bool)
    nlp = pc.oracle_nlp(name, **over)
    x, p = mpc.opt_x_num.master.copy(), mpc.opt_p_num.master.copy()
    J = nd.constraint_jacobian()
    Jo = sps.csr_matrix(nlp.jac(x, p))
    used = np.ones(mpc.structure.n_opt_x, bool)
    used[mpc.structure.tables["dummy_idx"]] = False
    dJ = (J - Jo)[:, used]
    assert abs(dJ).max() < 1e-6 * max(1.0, abs(Jo).max()), abs(dJ).max()
    # the active set is the reference's: every equality row, the bounds the solution sits on
    g_in, x_in, g_act, x_act = active_constraints(x, mpc.opt_g_num, mpc._lb_opt_x.master, mpc._ub_opt_x.master, mpc._nlp_cons_lb,
                                                  mpc._nlp_cons_ub, nd.settings.active_set_tol)
    assert set(np.where(mpc._nlp_cons_lb == mpc._nlp_cons_ub)[0]) <= set(g_act)
    assert st.n_active_g == g_act.size and st.n_active_x == x_act.size
    return mpc, nd


def check_active_set_reduction(make_mpc, name, **over):
    """settings.active_set_reduction against the reference's algorithm restated with the oracle's functions (_nlpdifferentiator.py:
    287-301, 430-468): KKT matrix of L = f + lam_g' g + lam_x' x over z = (x, lam_g, lam_x), rows and columns of the inactive
    constraints removed, A S = -B solved with a sparse LU; x0 and u_prev columns (they enter linearly: B from a difference)."""
    from do_mpc_amd.differentiator import active_constraints
    mpc = solved(make_mpc, name, **over)
    nd = DoMPCDifferentiator(mpc, active_set_reduction=True)
    dxdp, _ = nd.differentiate()
    assert nd.status.reduced_nlp
    nlp = pc.oracle_nlp(name, **over)
    x, p0, lam = mpc.opt_x_num.master.copy(), mpc.opt_p_num.master.copy(), np.asarray(mpc.lam_g_num, float)
    n, m = x.size, lam.size
    _, _, g_act, x_act = active_constraints(x, mpc.opt_g_num, mpc._lb_opt_x.master, mpc._ub_opt_x.master, mpc._nlp_cons_lb,
                                            mpc._nlp_cons_ub, nd.settings.active_set_tol)
    W, A = sps.csr_matrix(nlp.hess(x, p0, 1.0, lam)), sps.csr_matrix(nlp.jac(x, p0))
    dummy = np.asarray(mpc.structure.tables["dummy_idx"])
    free_dummy = np.setdiff1d(dummy, x_act)                # unused variables without an active bound: the reference removes them from the NLP
    pin = np.zeros(n)
    pin[free_dummy] = 1.0
    E = sps.csr_matrix((np.ones(x_act.size), (np.arange(x_act.size), x_act)), shape=(x_act.size, n))
    Aa = A[g_act]
    K = sps.bmat([[W + sps.diags(pin), Aa.T, E.T], [Aa, None, None], [E, None, None]], format="csc")
    lu = spla.splu(K)
    lay = mpc._opt_p_layout
    cols = np.concatenate([lay.resolve(("_x0",)).ravel(), lay.resolve(("_u_prev",)).ravel()])

    def grad_z(p):
        return np.concatenate([nlp.grad(x, p) + nlp.jac(x, p).T @ lam, nlp.g(x, p)[g_act], np.zeros(x_act.size)])

    g0 = grad_z(p0)
    ref = np.zeros((n, len(cols)))
    for k, j in enumerate(cols):
        p = p0.copy()
        h = max(1.0, abs(p0[j]))
        p[j] += h
        rhs = -(grad_z(p) - g0) / h
        sol = lu.solve(rhs)
        sol += lu.solve(rhs - K @ sol)
        ref[:, k] = sol[:n]
    ref *= mpc.opt_x_scaling.master[:, None]
    used = np.ones(n, bool)
    used[dummy] = False
    got = np.asarray(dxdp)[used][:, cols]
    err = np.max(np.abs(got - ref[used]))
    assert err < 1e-8 * max(1.0, np.max(np.abs(ref[used]))), err          # (measured 5e-11 / 7e-15 / 8e-12 on the host emulation)
    # the variables held by an active bound do not move
    if x_act.size:
        assert np.max(np.abs(np.asarray(dxdp)[x_act][:, cols])) < 1e-7 * max(1.0, np.max(np.abs(ref)))
    return err


def check_standalone_nlp_differentiator():
    """NLPDifferentiator on the example of the reference's class docstring (_nlpdifferentiator.py:27-62: a Rosenbrock-like objective,
    two inequality rows that depend on the parameter) against finite differences of re-solves (scipy SLSQP), plus the bookkeeping:
    unused variables / parameters are removed and come back as zero rows / columns."""
    from scipy.optimize import minimize
    from do_mpc_amd import sym
    from do_mpc_amd.differentiator import NLPDifferentiator
    x, p = sym.SX.sym("x", 3), sym.SX.sym("p", 2)            # x[2] and p[1] appear nowhere
    f = (1 - x[0]) ** 2 + 0.2 * (x[1] - x[0] ** 2) ** 2
    ci = (x[0] + 0.5) ** 2 + x[1] ** 2
    g = sym.vertcat(p[0] ** 2 / 4 - ci, ci - p[0] ** 2)
    bounds = {"lbx": np.array([0, -np.inf, -np.inf]), "ubx": np.full(3, np.inf), "lbg": np.full(2, -np.inf), "ubg": np.zeros(2)}
    nd = NLPDifferentiator({"x": x, "p": p, "f": f, "g": g}, bounds, check_rank=True)
    assert nd.status.reduced_nlp and nd.status.sym_KKT and nd.n_x == 2 and nd.n_p == 1
    F = sym.Function("F", [x, p], [f, g, sym.gradient(f, x), sym.jacobian(g, x)])

    def solve(pv):
        r = minimize(lambda v: F.eval(np.append(v, 0.0), [pv, 0.0])[0][0], [0.5, 0.5], method="SLSQP", bounds=[(0, None), (None, None)],
                     constraints=[{"type": "ineq", "fun": lambda v: -F.eval(np.append(v, 0.0), [pv, 0.0])[1]}],
                     options={"ftol": 1e-15, "maxiter": 500})
        return np.append(r.x, 0.0)

    p0 = 1.0
    xs = solve(p0)
    _, gv, gf, Jg = F.eval(xs, [p0, 0.0])
    Jg = Jg.reshape(2, 3, order="F")
    act = np.abs(gv) < 1e-6
    assert act.tolist() == [False, True]
    lam_g = np.zeros(2)
    lam_g[act] = np.linalg.lstsq(Jg[act][:, :2].T, -gf[:2], rcond=None)[0]
    sol = {"x": xs, "g": gv, "lam_g": lam_g, "lam_x": np.zeros(3)}
    dx, dl = nd.differentiate(sol, np.array([p0, 0.0]))
    assert nd.status.LICQ and nd.status.SC and nd.status.lse_solved and nd.status.full_rank and nd.status.residuals < 1e-10
    h = 1e-5
    fd = (solve(p0 + h) - solve(p0 - h)) / (2 * h)
    assert dx.shape == (3, 2) and dl.shape == (2 + 3, 2)
    assert np.max(np.abs(np.asarray(dx)[:, 0] - fd)) < 5e-4 * np.max(np.abs(fd)), (dx, fd)
    assert np.all(np.asarray(dx)[2] == 0) and np.all(np.asarray(dx)[:, 1] == 0) and np.all(np.asarray(dl)[0] == 0)   # unused / inactive
    # a solution at which strict complementarity fails is reported, not hidden
    sol2 = dict(sol, lam_g=np.zeros(2))
    nd.differentiate(sol2, np.array([p0, 0.0]))
    assert nd.status.SC is False
    for bad in ({"x": xs}, 3):
        try:
            nd.differentiate(bad, np.array([p0, 0.0]))
        except ValueError:
            pass
        else:
            raise AssertionError("accepted an incomplete nlp_sol")


def check_singular_reduced_system_is_reported(make_mpc):
    """industrial_poly: without the barrier terms of its inactive bounds the reduced KKT system is singular along the flat direction
    of the problem (three weakly determined entries) - the reference's dense solve would return NaNs
    (_solve_linear_system); here the structured factorisation reports the wrong inertia instead of returning numbers."""
    mpc = solved(make_mpc, "industrial_poly")
    dxdp, _ = DoMPCDifferentiator(mpc).differentiate()                    # the barrier problem's sensitivities exist
    assert dxdp.shape == (mpc.structure.n_opt_x, mpc.structure.n_opt_p)
    nd = DoMPCDifferentiator(mpc, active_set_reduction=True)
    try:
        nd.differentiate()
    except RuntimeError as e:
        assert "wrong inertia" in str(e)
    else:
        raise AssertionError("a singular reduced system went unnoticed")
    assert not nd.status.lse_solved and nd.status.reduced_nlp
