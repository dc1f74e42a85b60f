#!/usr/bin/env python3
"""Reproduce the flaky 37-problems-on-4-slots batch: first some industrial_poly work in the same process, then repeats."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from do_mpc_amd.examples import CASES
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if pre:
    exi = CASES["industrial_poly"]
    m = exi.build_mpc(exi.build_model(), max_batch=12)
    X0 = bench.synthetic_x0_batch(12)
    r = m.make_step_batch(X0); r2 = m.make_step_batch(X0)
    print("pre ok", bool(r["stats"]["success"].all()), flush=True)
    m1 = exi.build_mpc(exi.build_model()); m1.x0 = X0[0]; m1.set_initial_guess(); m1.make_step(X0[0])
    import gc
    del m, m1, r, r2
    gc.collect()
ex = CASES["batch_reactor"]
for rep in range(reps):
    mpc = ex.build_mpc(ex.build_model(), max_batch=4, nlpsol_opts={"ipopt.max_iter": 150})
    rng = np.random.default_rng(3)
    X0 = ex.X0 * (1 + 0.05 * rng.uniform(-1, 1, size=(37, 4)))
    ps = mpc.structure
    P = np.tile(mpc.opt_p_num.master, (37, 1)); P[:, :4] = X0
    P[:, ps.p_off_p:ps.p_off_uprev] = mpc.p_fun(0.0).master
    Xi = np.zeros((37, ps.n_opt_x)); Xi[:, :ps.off_z].reshape(37, -1, 4)[:] = X0[:, None, :]
    r = mpc.S.solve_batch(Xi, mpc._lb_opt_x.master, mpc._ub_opt_x.master, mpc._nlp_cons_lb, mpc._nlp_cons_ub, P)
    st = r["stats"]
    bad = np.where(st["success"] == 0)[0]
    print(rep, "fail:", bad, "status", st["status"][bad], "iters", st["iter_count"][bad], "inf_pr", st["inf_pr"][bad], "obj", st["obj"][bad], flush=True)
    del mpc, r
    import gc; gc.collect()
