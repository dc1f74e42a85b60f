"""Stress and hardening tests of the HIP path (run on the MI355X with -m gpu):

* the mixed-model scenario in which a timing-dependent failure was once seen (37 batch_reactor problems on
  4 slots after industrial_poly work in the same process), repeated, with bitwise comparison of the repetitions;
* the wide mode (K workgroups per problem, device-scope barrier) on a non-convex problem that needs inertia correction
  in most iterations - the factorisation is repeated and its verdict crosses the workgroups through the shared flags;
* the stop request / watchdog of the blocking entry points.
"""
import gc

import numpy as np
import pytest

import parity_common as pc
from do_mpc_amd.examples import CASES

pytestmark = pytest.mark.gpu


def _batch_reactor_inputs(mpc, n=37):
    ex = CASES["batch_reactor"]
    rng = np.random.default_rng(3)
    X0 = ex.X0 * (1 + 0.05 * rng.uniform(-1, 1, size=(n, 4)))
    ps = mpc.structure
    P = np.tile(mpc.opt_p_num.master, (n, 1))
    P[:, :4] = X0
    P[:, ps.p_off_p:ps.p_off_uprev] = mpc.p_fun(0.0).master
    Xi = np.zeros((n, ps.n_opt_x))
    Xi[:, :ps.off_z].reshape(n, -1, 4)[:] = X0[:, None, :]
    return Xi, P


def test_mixed_model_batches_repeat_bitwise():
    import bench
    exi = CASES["industrial_poly"]
    ref = None
    for rep in range(4):
        m = exi.build_mpc(exi.build_model(), max_batch=12)
        X0 = bench.synthetic_x0_batch(12)
        r = m.make_step_batch(X0)
        assert r["stats"]["success"].all()
        m1 = exi.build_mpc(exi.build_model())
        m1.x0 = X0[0]
        m1.set_initial_guess()
        m1.make_step(X0[0])
        del m, m1, r
        gc.collect()
        ex = CASES["batch_reactor"]
        mpc = ex.build_mpc(ex.build_model(), max_batch=4, nlpsol_opts={"ipopt.max_iter": 150})     # 4 slots, 37 problems
        Xi, P = _batch_reactor_inputs(mpc)
        r = mpc.S.solve_batch(Xi, mpc._lb_opt_x.master, mpc._ub_opt_x.master, mpc._nlp_cons_lb, mpc._nlp_cons_ub, P)
        st = r["stats"]
        assert st["success"].all(), (rep, st[st["success"] == 0])
        assert st["n_ls_fail"].max() == 0
        if ref is None:
            ref = (r["x"].copy(), st["iter_count"].copy())
        else:
            assert np.array_equal(st["iter_count"], ref[1]), rep
            assert np.array_equal(r["x"], ref[0]), rep
        del mpc, r
        gc.collect()


@pytest.mark.parametrize("block", [64, 128])
def test_fewer_wavefronts_per_problem_give_the_same_solution(block):
    """block_threads = 64 / 128: one or two wavefronts per problem and 4x / 2x the resident problem slots (large
    batches); only the grouping of the reductions differs from the 256-thread workgroup."""
    name = "batch_reactor"
    ex = CASES[name]
    ref = ex.build_mpc(ex.build_model(), max_batch=40)
    Xi, P = _batch_reactor_inputs(ref, 40)
    args = (Xi, ref._lb_opt_x.master, ref._ub_opt_x.master, ref._nlp_cons_lb, ref._nlp_cons_ub, P)
    r0 = ref.S.solve_batch(*args)
    mpc = ex.build_mpc(ex.build_model(), max_batch=40, block_threads=block)
    r1 = mpc.S.solve_batch(*args)
    assert r0["stats"]["success"].all() and r1["stats"]["success"].all()
    assert np.abs(r1["stats"]["iter_count"] - r0["stats"]["iter_count"]).max() <= 1
    iu = ref.structure.iu(0, 0)
    assert pc.relerr(r1["x"][:, iu], r0["x"][:, iu]) < 1e-8


def test_wide_mode_stress_with_inertia_correction(monkeypatch):
    ex = CASES["CSTR"]
    kw = dict(track_sign=-3000.0)          # concave cost: ~2/3 of the iterations repeat the factorisation with delta_w > 0
    res = {}
    for K in ("1", "4", "32"):
        monkeypatch.setenv("DOMPC_WIDE", K)
        runs = []
        for rep in range(3):
            mpc = ex.build_mpc(ex.build_model(), **kw)
            mpc.x0 = ex.X0
            mpc.set_initial_guess()
            u0 = mpc.make_step(ex.X0).ravel().copy()
            st = dict(mpc.solver_stats)
            assert st["success"], (K, rep, st)
            assert st["n_reg"] > 10, st
            runs.append((u0, st["iter_count"], st["n_reg"], mpc.opt_x_num.master.copy()))
        for r in runs[1:]:                 # the same launch shape is bitwise reproducible
            assert r[1] == runs[0][1] and r[2] == runs[0][2] and np.array_equal(r[3], runs[0][3]), K
        res[K] = runs[0]
    for K in ("4", "32"):                  # other reduction grouping: same local solution, same path up to rounding
        assert pc.relerr(res[K][0], res["1"][0]) < 1e-6, (K, res[K][0], res["1"][0])
        assert abs(res[K][1] - res["1"][1]) <= 3


def test_stop_request_and_watchdog(monkeypatch):
    import bench
    exi = CASES["industrial_poly"]
    X0 = bench.synthetic_x0_batch(8)
    mpc = exi.build_mpc(exi.build_model(), max_batch=8)
    mpc.S.abort(True)                      # raised before the call: every problem leaves at its first check
    r = mpc.make_step_batch(X0)
    assert (r["stats"]["status"] == 6).all() and not r["stats"]["success"].any()
    mpc.S.abort(False)
    r = mpc.make_step_batch(X0)
    assert r["stats"]["success"].all()
    # watchdog of the blocking call: 1 ms is far less than these solves take -> stop request, statuses 6, no hang
    monkeypatch.setenv("DOMPC_WATCHDOG_S", "0.001")
    m2 = exi.build_mpc(exi.build_model(), max_batch=8)
    r2 = m2.make_step_batch(X0)
    assert (r2["stats"]["status"] == 6).any()
    monkeypatch.delenv("DOMPC_WATCHDOG_S")
    r3 = m2.make_step_batch(X0)            # the handle is re-armed after the watchdog fired
    assert r3["stats"]["success"].all() or (r3["stats"]["status"] == 6).any()
