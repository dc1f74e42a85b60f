"""Latency of small batches (wide mode: several workgroups per problem): cold make_step of industrial_poly, B = 1 / 8 / 64.
Usage: python tools/gpu_b1.py [case]      (DOMPC_DEFS selects a measurement build, DOMPC_WIDE the workgroups per problem)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from do_mpc_amd.examples import CASES

name = sys.argv[1] if len(sys.argv) > 1 else "industrial_poly"
ex = CASES[name]
mpc = ex.build_mpc(ex.build_model())
x0 = ex.X0
ts, its, u = [], [], None
for rep in range(7):
    mpc.reset_history() if hasattr(mpc, "reset_history") else None
    mpc.x0 = x0
    mpc.set_initial_guess()
    t = time.perf_counter()
    u = mpc.make_step(x0).ravel()
    ts.append((time.perf_counter() - t) * 1e3)
    its.append(mpc.solver_stats["iter_count"])
    assert mpc.solver_stats["success"]
print(f"{name} B=1 cold: min {min(ts[1:]):.2f} ms  median {np.median(ts[1:]):.2f} ms  iters {its[-1]}  kernel {mpc.solver_stats['t_wall_total']*1e3:.2f} ms  u0 {u!r}")
for B in (8, 64):
    rng = np.random.default_rng(0)
    X0 = x0[None, :] * (1.0 + 0.01 * rng.standard_normal((B, x0.size))) if name != "industrial_poly" else np.tile(x0, (B, 1))
    best = 1e9
    for rep in range(4):
        t = time.perf_counter()
        r = mpc.make_step_batch(X0)
        best = min(best, (time.perf_counter() - t) * 1e3)
    st = r["stats"]
    print(f"{name} B={B} cold: {best:.2f} ms  converged {int(np.sum(st['success']))}  u0[0] {r['u0'][0]!r}")
