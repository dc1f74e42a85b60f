"""Wide mode with eight wavefronts per workgroup (512 threads; code object built with -DDOMPC_MAXBLOCK=512) against four, for the
243-leaf tree and the 9-scenario problem:  python tools/gpu_wide_block.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DOMPC_DEFS"] = "DOMPC_MAXBLOCK=512"
import numpy as np
from do_mpc_amd.examples import industrial_poly as ex


def run(mpc, label, reps=3):
    ts, its, u0 = [], [], None
    for k in range(reps):
        mpc.x0 = ex.X0; mpc.u0 = np.zeros(3); mpc._t0 = mpc._t0 * 0; mpc.set_initial_guess()
        t = time.perf_counter(); u0 = mpc.make_step(ex.X0); ts.append((time.perf_counter() - t) * 1e3)
        its.append(mpc.solver_stats["iter_count"])
    print("%-34s best %.1f ms  (all %s)  it=%s %s u0=%s" % (label, min(ts), " ".join("%.1f" % t for t in ts), its[-1],
                                                          mpc.solver_stats["return_status"], np.array2string(u0.ravel(), precision=10)), flush=True)


mpc = ex.build_mpc(ex.build_model(), n_robust=5, uncertainty="paired")
for blk, K in ((256, 112), (512, 32), (512, 48), (512, 56), (512, 64), (512, 80), (512, 96), (512, 112), (512, 128)):
    os.environ["DOMPC_WIDE"] = str(K); os.environ["DOMPC_WIDE_SPREAD"] = "1"; os.environ["DOMPC_WIDE_BLOCK"] = str(blk)
    run(mpc, "tree block=%d K=%d" % (blk, K))
del mpc
mpc = ex.build_mpc(ex.build_model())
for blk, K in ((256, 22), (512, 8), (512, 11), (512, 16), (512, 22)):
    os.environ["DOMPC_WIDE"] = str(K); os.environ["DOMPC_WIDE_SPREAD"] = "1"; os.environ["DOMPC_WIDE_BLOCK"] = str(blk)
    run(mpc, "9-scenario block=%d K=%d" % (blk, K), reps=4)
