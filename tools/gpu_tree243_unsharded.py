import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from do_mpc_amd.examples import industrial_poly as ex
mpc = ex.build_mpc(ex.build_model(), n_robust=5, uncertainty="paired")
for k in range(3):
    mpc.x0 = ex.X0; mpc.u0 = np.zeros(3); mpc._t0 = mpc._t0 * 0; mpc.set_initial_guess()
    t = time.perf_counter(); u0 = mpc.make_step(ex.X0); dt = time.perf_counter() - t
    print("unsharded 243-leaf: %.1f ms, it=%d %s u0=%s" % (dt * 1e3, mpc.solver_stats["iter_count"], mpc.solver_stats["return_status"], u0.ravel()), flush=True)
