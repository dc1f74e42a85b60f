"""Iteration trace (it, mu, E0, inf_pr, inf_du, alpha, delta_w, obj) of one member of the timed batch, B = 1 (wide mode) and inside a
B = 4096 launch (one wavefront per problem); saved to gpurun_out/trace_member_<i>.npy.   python tools/gpu_trace_member.py <member>"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from do_mpc_amd.examples import industrial_poly as ex
X0 = bench.synthetic_x0_batch(16384)
i = int(sys.argv[1])
for B in (1, 4096):
    mpc = ex.build_mpc(ex.build_model(), max_batch=B)
    r = mpc.make_step_batch(np.tile(X0[i], (B, 1)))
    n = int(r["stats"]["iter_count"][0]) + 1
    tr = np.asarray(mpc.S.trace(4096))[:n]
    print("B", B, "iterations", r["stats"]["iter_count"][:2], "status", r["stats"]["success"][:2])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", f"trace_member_{i}_B{B}.npy"), tr)
