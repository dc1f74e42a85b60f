"""One problem alone on the whole chip: workgroups per problem (K) and their placement (one XCD / all XCDs) for the 243-leaf
industrial_poly tree (BASELINE configs[4], 4 008 edges, unsharded) and for the shipped 9-scenario problem (180 edges).
    python tools/gpu_wide_spread.py [tree|b1|both]
DOMPC_WIDE / DOMPC_WIDE_SPREAD are read by the runtime at every call (dompc_solve_batch_device)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from do_mpc_amd.examples import industrial_poly as ex

what = sys.argv[1] if len(sys.argv) > 1 else "both"


def run(mpc, label, reps=3):
    ts, its, u0 = [], [], None
    for k in range(reps):
        mpc.x0 = ex.X0; mpc.u0 = np.zeros(3); mpc._t0 = mpc._t0 * 0; mpc.set_initial_guess()
        t = time.perf_counter(); u0 = mpc.make_step(ex.X0); ts.append((time.perf_counter() - t) * 1e3)
        its.append(mpc.solver_stats["iter_count"])
    print("%-28s best %.1f ms  (all %s)  it=%s %s u0=%s" % (label, min(ts), " ".join("%.1f" % t for t in ts), its[-1],
                                                          mpc.solver_stats["return_status"], np.array2string(u0.ravel(), precision=10)), flush=True)


if what in ("tree", "both"):
    mpc = ex.build_mpc(ex.build_model(), n_robust=5, uncertainty="paired")
    print("243-leaf tree: edges", mpc.structure.n_edges, flush=True)
    for spread, K in ((0, 16), (0, 32), (1, 32), (1, 64), (1, 96), (1, 128), (1, 192), (1, 256)):
        os.environ["DOMPC_WIDE"] = str(K); os.environ["DOMPC_WIDE_SPREAD"] = str(spread)
        run(mpc, "tree spread=%d K=%d" % (spread, K))
    os.environ.pop("DOMPC_WIDE"); os.environ.pop("DOMPC_WIDE_SPREAD")
    run(mpc, "tree default rule")
    del mpc
if what in ("b1", "both"):
    mpc = ex.build_mpc(ex.build_model())
    for spread, K in ((0, 16), (1, 8), (1, 16), (1, 24), (1, 32), (1, 48)):
        os.environ["DOMPC_WIDE"] = str(K); os.environ["DOMPC_WIDE_SPREAD"] = str(spread)
        run(mpc, "9-scenario spread=%d K=%d" % (spread, K), reps=4)
