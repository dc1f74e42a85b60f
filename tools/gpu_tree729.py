"""One size beyond BASELINE configs[4]: industrial_poly with 3^6 = 729 leaves (656 100 variables, 12 024 edges) on ONE GPU in the
whole-chip wide mode - where does a single MI355X stop being enough?   python tools/gpu_tree729.py [n_robust]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from do_mpc_amd.examples import industrial_poly as ex
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 6
t = time.perf_counter()
mpc = ex.build_mpc(ex.build_model(), n_robust=nr, uncertainty="paired")
ps = mpc.structure
print("leaves %d, n_opt_x %d, n_g %d, edges %d, setup %.1f s" % (ps.S, ps.n_opt_x, ps.n_g, ps.n_edges, time.perf_counter() - t), flush=True)
for K in (None, 128, 192, 256):
    if K is None: os.environ.pop("DOMPC_WIDE", None)
    else: os.environ["DOMPC_WIDE"] = str(K)
    ts = []
    for k in range(3):
        mpc.x0 = ex.X0; mpc.u0 = np.zeros(3); mpc._t0 = mpc._t0 * 0; mpc.set_initial_guess()
        t = time.perf_counter(); u0 = mpc.make_step(ex.X0); ts.append((time.perf_counter() - t) * 1e3)
    st = mpc.solver_stats
    print("K=%s best %.1f ms (all %s) it=%d %s u0=%s  -> %.2f ms per iteration" % (K, min(ts), " ".join("%.1f" % v for v in ts), st["iter_count"], st["return_status"],
          np.array2string(u0.ravel(), precision=8), min(ts) / max(st["iter_count"], 1)), flush=True)
