"""Edge cases of the hot path on the host emulation of the kernels: shortest horizons, a horizon shorter than the
branching depth allows, empty batches, one-problem batches, and repeated solves on one handle."""
import numpy as np
import pytest

import hostemu
import parity_common as pc
from do_mpc_amd.examples import CASES
from oracle import ipm


def make(name, **kw):
    ex = CASES[name]
    with hostemu.patched():
        return ex.build_mpc(ex.build_model(), **kw)


@pytest.mark.parametrize("name,kw", [("batch_reactor", {"n_horizon": 1}), ("batch_reactor", {"n_horizon": 2}),
                                      ("CSTR", {"n_horizon": 1, "n_robust": 1}), ("CSTR", {"n_horizon": 3, "n_robust": 0}),
                                      ("oscillating_masses", {"n_horizon": 1})])
def test_short_horizons_match_the_oracle(name, kw):
    """One- to three-stage problems are nearly flat in some input directions (the optimum of the CSTR moves by
    3e-3 between mu = 1e-9 and mu = 0, see parity_common.py), so the comparison is on what is determined: the
    optimal cost (1e-7 relative), the KKT conditions evaluated with the oracle's functions, and u0 at 1e-3."""
    ex = CASES[name]
    mpc = make(name, **kw)
    nlp = pc.oracle_nlp(name, **kw)
    assert (nlp.n_opt_x, nlp.n_g) == (mpc.structure.n_opt_x, mpc.structure.n_g)
    mpc.x0 = ex.X0
    mpc.set_initial_guess()
    u0 = mpc.make_step(ex.X0).ravel()
    assert mpc.solver_stats["success"], mpc.solver_stats
    p = nlp.opt_p(ex.X0, np.zeros(nlp.nu))
    r = ipm.solve(nlp, nlp.initial_guess(ex.X0), p)
    assert r["stats"]["success"]
    f_ours, f_ref = nlp.f(mpc.opt_x_num.master, p), nlp.f(r["x"], p)
    assert abs(f_ours - f_ref) <= 1e-7 * max(1.0, abs(f_ref)), (f_ours, f_ref)
    assert pc.relerr(u0, nlp.u0_of(r["x"])) < 1e-3
    pc.check_kkt_with_oracle_functions(mpc, nlp, ex.X0)


def test_n_robust_larger_than_horizon_is_refused_like_the_reference():
    with pytest.raises(Exception, match="n_robust"):
        make("CSTR", n_horizon=2, n_robust=3)


def test_empty_and_single_batches():
    ex = CASES["batch_reactor"]
    mpc = make("batch_reactor", max_batch=4)
    ps = mpc.structure
    r0 = mpc.S.solve_batch(np.zeros((0, ps.n_opt_x)), mpc._lb_opt_x.master, mpc._ub_opt_x.master, mpc._nlp_cons_lb,
                           mpc._nlp_cons_ub, np.zeros((0, ps.n_opt_p)))
    assert r0["x"].shape == (0, ps.n_opt_x) and r0["stats"].shape == (0,)
    r1 = mpc.make_step_batch(ex.X0[None, :])
    mpc.x0 = ex.X0
    mpc.set_initial_guess()
    u = mpc.make_step(ex.X0).ravel()
    assert r1["stats"]["success"].all() and pc.relerr(r1["u0"][0], u) < 1e-12


def test_repeated_solves_on_one_handle_are_reproducible():
    ex = CASES["CSTR"]
    mpc = make("CSTR")
    out = []
    for _ in range(3):
        mpc.x0 = ex.X0
        mpc.u0 = np.zeros(2)
        mpc._t0 = mpc._t0 * 0
        mpc.set_initial_guess()
        out.append((mpc.make_step(ex.X0).ravel().copy(), mpc.opt_x_num.master.copy(), mpc.solver_stats["iter_count"]))
    for u, x, it in out[1:]:
        assert np.array_equal(u, out[0][0]) and np.array_equal(x, out[0][1]) and it == out[0][2]
