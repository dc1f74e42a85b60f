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
