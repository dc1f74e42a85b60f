"""One-off check on the GPU box: the docstring's cost terms on a continuous model send the model through the dense edge path - here as
members of a batch launch (one wavefront per problem), against the single solve.  python tools/gpu_route_dense_batch.py [case] [B]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import route_cases as rc
from do_mpc_amd.examples import CASES
name = sys.argv[1] if len(sys.argv) > 1 else "industrial_poly"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ex = CASES[name]
mpc = rc.stopped_before_setup(lambda n: ex.build_mpc(ex.build_model()), name)
mpc.settings.max_batch = B
mpc.prepare_nlp()
rc.ADDED_COST["docstring"](mpc)
t0 = time.time(); mpc.create_nlp(); print("create_nlp %.1f s" % (time.time() - t0), flush=True)
x0 = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))["mpc._x"][0]
mpc.x0 = x0; mpc.set_initial_guess(); mpc.make_step(x0)
print("single:", mpc.solver_stats["iter_count"], mpc.solver_stats["success"], flush=True)
t0 = time.time(); r = mpc.make_step_batch(np.tile(x0, (B, 1))); dt = time.time() - t0
print("batch %d: %.2f s, converged %d, iters %s, max |x - single| %.2e" % (B, dt, int(np.sum(r["stats"]["success"])), np.unique(r["stats"]["iter_count"]),
      np.max(np.abs(r["x"] - mpc.opt_x_num.master))))
