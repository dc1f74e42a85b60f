import os, sys, time
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import route_cases as rc
from do_mpc_amd.examples import CASES
B = int(sys.argv[1])
name = "industrial_poly"; ex = CASES[name]
rng = np.random.default_rng(0)
X0 = np.asarray(ex.X0, float)[None, :] * (1.0 + 0.002 * rng.standard_normal((B, len(ex.X0))))
mpc = rc.stopped_before_setup(lambda n: ex.build_mpc(ex.build_model()), name)
mpc.settings.max_batch = B
mpc.prepare_nlp(); rc.rows_at_three_nodes(mpc, name); mpc.create_nlp()
for rep in range(3):
    t = time.perf_counter(); r = mpc.make_step_batch(X0); print("total %.3f" % (time.perf_counter() - t), flush=True)
