"""Cold batch of one example: time per launch, steps/s, iterations, u0 of problem 0.  python tools/gpu_time_case.py batch_reactor '{"n_horizon": 50}' 16384
(DOMPC_DEFS selects a measurement build)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from do_mpc_amd.examples import CASES
name, kw, B = sys.argv[1], json.loads(sys.argv[2]), int(sys.argv[3])
ex = CASES[name]
mpc = ex.build_mpc(ex.build_model(), max_batch=B, **kw)
rng = np.random.default_rng(0)
X0 = np.asarray(ex.X0, float)[None, :] * (1.0 + 0.01 * rng.standard_normal((B, len(ex.X0))))
best = 1e9
for rep in range(4):
    t = time.perf_counter()
    r = mpc.make_step_batch(X0)
    best = min(best, time.perf_counter() - t)
st = r["stats"]
print("[%s] %s B=%d: %.2f ms  %.0f steps/s  converged %d  iterations mean %.3f  u0[0] %r  u0[-1] %r" % (
    os.environ.get("DOMPC_DEFS", ""), name, B, best * 1e3, B / best, int(np.sum(st["success"])), float(np.mean(st["iter_count"])), r["u0"][0], r["u0"][-1]))
