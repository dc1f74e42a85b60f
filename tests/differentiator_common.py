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
