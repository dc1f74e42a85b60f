#!/usr/bin/env python3
"""Wide-mode smoke: single solves of each case with the given DOMPC_WIDE (workgroups per problem)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from do_mpc_amd.examples import CASES
name = sys.argv[1]
opts = {}
if len(sys.argv) > 2:
    opts = {"ipopt.max_iter": int(sys.argv[2])}
ex = CASES[name]
mpc = ex.build_mpc(ex.build_model(), nlpsol_opts=opts)
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
mpc.x0 = ex.X0; mpc.set_initial_guess()
t = time.time(); u0 = mpc.make_step(g["mpc._x"][0]).ravel(); dt = time.time() - t
print(name, "K", os.environ.get("DOMPC_WIDE"), mpc.solver_stats["return_status"], mpc.solver_stats["iter_count"], f"{dt*1e3:.1f} ms", u0, g["mpc._u"][0], flush=True)
