"""Wall time of DoMPCDifferentiator.differentiate() (all sensitivity columns in one batched launch) after a converged make_step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from do_mpc_amd.examples import CASES
from do_mpc_amd.differentiator import DoMPCDifferentiator
for name in ("batch_reactor", "industrial_poly"):
    ex = CASES[name]
    mpc = ex.build_mpc(ex.build_model(), max_batch=128)
    mpc.x0 = ex.X0; mpc.set_initial_guess(); mpc.make_step(ex.X0)
    d = DoMPCDifferentiator(mpc)
    d.differentiate()
    t = time.perf_counter(); dx, dl = d.differentiate(); dt = (time.perf_counter() - t) * 1e3
    print(f"{name}: n_x {d.n_x} n_p {d.n_p} newton solves {d.status['n_newton_solves']}  differentiate() {dt:.1f} ms  |dxdp|max {np.abs(dx).max():.3e}")
