"""What the modifications of the low-level route cost in a batch launch (GPU): industrial_poly, B cold solves, plain / + cost terms / + rows /
+ the docstring's terms in the collocation states (dense edge path).  python tools/gpu_route_cost.py [B]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import route_cases as rc
from do_mpc_amd.examples import CASES
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
name = "industrial_poly"
ex = CASES[name]
rng = np.random.default_rng(0)
X0 = np.asarray(ex.X0, float)[None, :] * (1.0 + 0.002 * rng.standard_normal((B, len(ex.X0))))
for label, cost, rows in (("plain", None, False), ("+ cost terms (leaves, inner node, root)", "tree", False), ("+ four rows at three nodes", None, True),
                          ("+ both", "tree", True), ("+ docstring terms in the collocation states (dense edge path)", "docstring", False)):
    mpc = rc.stopped_before_setup(lambda n: ex.build_mpc(ex.build_model()), name)
    mpc.settings.max_batch = B
    mpc.prepare_nlp()
    if rows:
        rc.rows_at_three_nodes(mpc, name)
    if cost:
        rc.ADDED_COST[cost](mpc)
    mpc.create_nlp()
    best = 1e9
    for rep in range(3):
        t = time.perf_counter(); r = mpc.make_step_batch(X0); best = min(best, time.perf_counter() - t)
    st = r["stats"]
    print("%-70s %8.1f ms  %7.0f steps/s  converged %d / %d  iterations %.2f" % (label, best * 1e3, B / best, int(np.sum(st["success"])), B, float(np.mean(st["iter_count"]))), flush=True)
    info = (mpc.S.inner if getattr(mpc.S, "row_mapped", False) else mpc.S).code_object_info
    print("      ", {k: (os.path.basename(v) if isinstance(v, str) and "/" in v else v) for k, v in info.items()}, "slots", mpc.S.num_slots, flush=True)
    mpc.S.close()
