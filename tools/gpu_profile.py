#!/usr/bin/env python3
"""Per-phase cycle breakdown of one solve (problem 0) from the kernel's own shader-clock counters."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DOMPC_PROFILE", "1")      # the code object with the sub-phase counters compiled in
import bench  # noqa: E402
from do_mpc_amd.examples import CASES  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "industrial_poly"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    kw = json.loads(sys.argv[3]) if len(sys.argv) > 3 else {}
    mhe = name.endswith(":mhe")            # the estimator of an example (rotating_masses:mhe): window 3 of the reference's stored run
    name = name.split(":")[0]
    ex = CASES[name]
    import time
    if mhe:
        est = ex.build_mhe(ex.build_model(), max_batch=max(B, 1), **kw)
        mpc = est._mpc
        g = np.load(os.path.join(ROOT, "tests", "golden", "rotating_masses.npz"))
        P = np.tile(np.asarray(g["estimator.opt_p_num"], float)[3][None, :], (B, 1))
        init = np.tile(np.asarray(g["estimator._opt_x_num"], float)[2][None, :], (B, 1))
        for rep in range(2):
            t = time.time()
            r = est.solve_batch(P, init)
            dt = time.time() - t
    else:
        mpc = ex.build_mpc(ex.build_model(), max_batch=max(B, 1), **kw)
        X0 = bench.synthetic_x0_batch(B) if name == "industrial_poly" else np.tile(ex.X0, (B, 1))
        for rep in range(2):
            t = time.time()
            r = mpc.make_step_batch(X0)
            dt = time.time() - t
    st = r["stats"]
    tr = mpc.S.trace(4096)[-1]
    tot = tr[5]
    print(f"{name} B={B}: wall {dt * 1e3:.1f} ms, iters {st['iter_count'][0]}, sweeps {st['n_sweeps'][0]}, trials {st['n_trials'][0]}")
    for nm, v in zip(("sweep", "riccati_bwd", "riccati_fwd", "linesearch", "measure"), tr[:5]):
        print(f"  {nm:12s} {v / 1e6:9.2f} Mcycles  {100 * v / tot:5.1f} %")
    for nm, v in zip(("step rules", "accept"), tr[6:8]):
        print(f"  {nm:12s} {v / 1e6:9.2f} Mcycles  {100 * v / tot:5.1f} %")
    print(f"  total        {tot / 1e6:9.2f} Mcycles (problem 0)")
    gb = mpc.S.trace(4096)[-6]
    if gb[5] > 0:
        print("  workgroup / device-scope barriers of problem 0: total %d = %.1f per iteration; sweep %d, backward %d, forward %d, line search %d, measure %d, step rules %d, accept %d"
              % (gb[5], gb[5] / max(1, st['iter_count'][0]), gb[0], gb[1], gb[2], gb[3], gb[4], gb[6], gb[7]))
    sub = np.concatenate([mpc.S.trace(4096)[-2][:8], mpc.S.trace(4096)[-3][:8], mpc.S.trace(4096)[-4][:8], mpc.S.trace(4096)[-5][:8]])
    names = {0: "edge:model-eval", 4: "edge:loads issued", 5: "edge:residual rows+H staging", 6: "edge:wait+build columns",
             1: "edge:dual pieces", 2: "edge:gauss-jordan+W", 7: "edge:tile condensing", 3: "edge:record stores",
             12: "node:staged tiles", 13: "node:own terms", 14: "node:own matrix", 8: "node:column to tile", 9: "node:coupling", 10: "node:cholesky+K", 11: "node:own part of the value function", 15: "node4:wait for the parent's operands + stores", 16: "fwd:node steps (chain walk)", 17: "fwd:edge loads issued",
             18: "fwd:edge wait + dw", 19: "fwd:edge rhs", 20: "fwd:edge dlam + stores", 21: "sweep:model evaluation", 22: "sweep:edge loop", 23: "sweep:node assembly",
             24: "node:closed loop (children, pass 2)", 25: "node:wait for the next node's operands", 26: "node:stores", 27: "chain:request of the next node's operands"}
    for i in (0, 4, 5, 6, 1, 2, 7, 3, 12, 13, 14, 8, 9, 10, 11, 24, 25, 26, 27, 15, 16, 17, 18, 19, 20, 21, 22, 23):
        print(f"    {names[i]:30s} {sub[i] / 1e6:9.2f} Mcycles")


if __name__ == "__main__":
    main()
