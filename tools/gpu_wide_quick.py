import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from do_mpc_amd.examples import industrial_poly as ex
def run(mpc, label, reps=4):
    ts=[]; 
    for k in range(reps):
        mpc.x0 = ex.X0; mpc.u0 = np.zeros(3); mpc._t0 = mpc._t0 * 0; mpc.set_initial_guess()
        t = time.perf_counter(); u0 = mpc.make_step(ex.X0); ts.append((time.perf_counter() - t) * 1e3)
    print("%-30s best %.1f ms (all %s) it=%d %s u0=%s" % (label, min(ts), " ".join("%.1f" % t for t in ts), mpc.solver_stats["iter_count"], mpc.solver_stats["return_status"], np.array2string(u0.ravel(), precision=10)), flush=True)
mpc = ex.build_mpc(ex.build_model(), n_robust=5, uncertainty="paired")
for K in (None, 64, 96, 128, 160, 192, 256):
    if K is None: os.environ.pop("DOMPC_WIDE", None)
    else: os.environ["DOMPC_WIDE"] = str(K)
    run(mpc, "tree K=%s" % K)
del mpc
mpc = ex.build_mpc(ex.build_model())
for K in (None, 16, 24, 32, 48):
    if K is None: os.environ.pop("DOMPC_WIDE", None)
    else: os.environ["DOMPC_WIDE"] = str(K)
    run(mpc, "9-scenario K=%s" % K)
