"""Pin the CPU oracle (oracle/) to the reference's own golden vectors.

Golden source: /root/reference/testing/results/results_*.pkl, asserted at 1e-8 by
/root/reference/testing/test_{industrial_poly,CSTR,batch_reactor,oscillating_masses_discrete}.py
and extracted to tests/golden/*.npz by tools/extract_golden.py.

Checks, per case:
  * NLP sizes (n_opt_x, n_g, n_opt_p) equal the reference's;
  * the golden primal solution satisfies our restated constraints (<= 2e-8) and, with the
    golden multipliers, stationarity on variables away from their bounds;
  * open-loop replay: feeding golden x[k], u[k-1] and warm-starting like
    Optimizer.solve (/root/reference/do_mpc/optimizer.py:754-768) reproduces golden u[k].
Tolerance for u (stated): 1e-6 relative to max(1,|u|) - the goldens are IPOPT iterates at
mu = 9.09e-10, and the CSTR optimum moves by 3e-3 relative between mu=1e-9 and mu=0, so
anything tighter would pin IPOPT's last Newton step rather than the solution.
"""
import os

import numpy as np
import pytest

from parity_common import golden_opt_p
from oracle import ipm
from oracle.models import CASES
from oracle.nlp import OracleNLP, collocation_coeffs
from oracle.nlp_dae import OracleNLPDae

GOLD = os.path.join(os.path.dirname(__file__), "golden")
U_RTOL = 1e-6

_cache = {}


def _nlp(name):
    if name not in _cache:
        case = CASES[name]()
        _cache[name] = (OracleNLPDae if case.get("z") else OracleNLP)(case)      # (`_z`: the general interval function, oracle/nlp_dae.py)
    return _cache[name]


def test_radau_coefficients():
    # /root/repo/SURVEY.md App. A.6 (validated against the goldens there)
    tau, C, D = collocation_coeffs(2, "radau")
    assert np.allclose(tau, [0, 1 / 3, 1])
    assert np.allclose(C, [[-4, -2, 2], [4.5, 1.5, -4.5], [-0.5, 0.5, 2.5]])
    assert np.allclose(D, [0, 0, 1])


@pytest.mark.parametrize("name", ["oscillating_masses", "batch_reactor", "CSTR", "industrial_poly", "rotating_masses",
                                  "oscillating_masses_dae", "dip"])
def test_golden_point_is_kkt_point_of_restated_nlp(name):
    nlp = _nlp(name)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    X, LG = g["mpc._opt_x_num"], g["mpc._lam_g_num"]
    P = np.stack([golden_opt_p(name, g, k, nlp.n_opt_p) for k in range(X.shape[0])])
    assert (nlp.n_opt_x, nlp.n_g, nlp.n_opt_p) == (X.shape[1], LG.shape[1], P.shape[1])
    s = nlp.scaling_vector()
    for k in range(X.shape[0]):
        x, p, lam = X[k] / s, P[k], LG[k]
        gv = nlp.g(x, p)
        eq = nlp.lbg == nlp.ubg
        assert np.max(np.abs(gv[eq])) < 2e-8
        if (~eq).any():
            assert np.max(gv[~eq] - nlp.ubg[~eq]) < 1e-7
        r = nlp.grad(x, p) + nlp.jac(x, p).T @ lam
        interior = (x - nlp.lbx > 1e-3 * np.maximum(1, np.abs(nlp.lbx))) & \
                   (nlp.ubx - x > 1e-3 * np.maximum(1, np.abs(nlp.ubx)))
        assert np.max(np.abs(r[interior])) < 2e-5 * max(1.0, np.max(np.abs(lam)))


@pytest.mark.parametrize("name,steps", [("oscillating_masses", 5), ("batch_reactor", 5), ("CSTR", 3),
                                        ("industrial_poly", 2), ("rotating_masses", 5), ("oscillating_masses_dae", 5)])
def test_open_loop_replay_matches_golden_u(name, steps):
    case = CASES[name]()
    nlp = _nlp(name)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    U, Xs, LG = g["mpc._u"], g["mpc._x"], g["mpc._lam_g_num"]
    xg = nlp.initial_guess(case["x0"])
    u_prev = np.zeros(nlp.nu)
    for k in range(steps):
        Pk = golden_opt_p(name, g, k, nlp.n_opt_p)
        tvp = Pk[nlp.p_off_tvp:nlp.p_off_p] if nlp.ntvp else None      # (the set-point staircase the reference read)
        p = nlp.opt_p(Xs[k], u_prev, tvp)
        assert np.allclose(p, Pk, rtol=0, atol=1e-12)
        r = ipm.solve(nlp, xg, p)
        assert r["stats"]["success"]
        u0 = nlp.u0_of(r["x"])
        assert np.max(np.abs(u0 - U[k]) / np.maximum(1.0, np.abs(U[k]))) < U_RTOL, (k, u0, U[k])
        # multipliers: same sign convention as CasADi (L = f + lam_g'g)
        assert np.max(np.abs(r["lam_g"] - LG[k])) < 1e-2 * max(1.0, np.max(np.abs(LG[k])))
        xg, u_prev = r["x"], U[k]


@pytest.mark.parametrize("name,steps", [("batch_reactor", 5), ("CSTR", 3)])
def test_with_ipopts_damping_of_one_sided_bounds_the_oracle_reproduces_the_goldens_to_rounding(name, steps):
    """kappa_d = 1e-5 (IPOPT's default: linear damping of the barrier for variables with one bound) was the last detail that
    separated the restated algorithm from IPOPT on these two cases: 1e-13 instead of 2e-11 (batch_reactor) / 1.4e-7 (CSTR)
    without it.  Asserted at 1e-10 on every variable that is in a constraint, u0 and the multipliers."""
    nlp = _nlp(name)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    s = nlp.scaling_vector()
    xg = nlp.initial_guess(g["mpc._x"][0])
    for k in range(steps):
        p = golden_opt_p(name, g, k, nlp.n_opt_p)
        r = ipm.solve(nlp, xg, p)
        assert r["stats"]["success"]
        X = g["mpc._opt_x_num"][k] / s
        used = np.diff(nlp.jac(X, p).tocsc().indptr) > 0
        assert np.max((np.abs(r["x"] - X) / np.maximum(1.0, np.abs(X)))[used]) < 1e-10
        assert np.max(np.abs(nlp.u0_of(r["x"]) - g["mpc._u"][k]) / np.maximum(1.0, np.abs(g["mpc._u"][k]))) < 1e-10
        LG = g["mpc._lam_g_num"][k]
        assert np.max(np.abs(r["lam_g"] - LG)) < 1e-9 * max(1.0, np.max(np.abs(LG)))
        xg = r["x"]


def test_dae_oracle_reproduces_the_discrete_dae_golden_to_rounding():
    """The general interval function (oracle/nlp_dae.py: algebraic rows, `_z` block) on the reference's discrete DAE example:
    primal solution incl. the algebraic states and every multiplier at 1e-12 (measured 1e-16) - and the same problem with the
    algebraic state substituted (oracle/nlp.py) gives the same inputs."""
    nlp, ode = _nlp("oscillating_masses_dae"), _nlp("oscillating_masses")
    g = np.load(os.path.join(GOLD, "oscillating_masses_dae.npz"))
    xg, u_prev = nlp.initial_guess(g["mpc._x"][0]), np.zeros(1)
    for k in range(5):
        p = nlp.opt_p(g["mpc._x"][k], u_prev)
        assert np.array_equal(p, g["mpc.opt_p_num"][k])
        r = ipm.solve(nlp, xg, p)
        assert r["stats"]["success"]
        assert np.max(np.abs(r["x"] * nlp.scaling_vector() - g["mpc._opt_x_num"][k])) < 1e-12
        assert np.max(np.abs(r["lam_g"] - g["mpc._lam_g_num"][k])) < 1e-12
        if k == 0:
            ro = ipm.solve(ode, ode.initial_guess(g["mpc._x"][0]), ode.opt_p(g["mpc._x"][0], u_prev))
            assert np.max(np.abs(ode.u0_of(ro["x"]) - nlp.u0_of(r["x"]))) < 1e-9
        xg, u_prev = r["x"], g["mpc._u"][k]
