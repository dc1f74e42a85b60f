"""Two-level barrier of the whole-chip wide mode (-DDOMPC_HIER_BARRIER=1) against the flat one: 243-leaf tree and the 9-scenario problem,
cold solves; the same bits are expected (the barrier does not touch the arithmetic).  python tools/gpu_hier_barrier.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from do_mpc_amd.examples import industrial_poly as ex


def run(mpc, label, reps=5):
    ts = []
    for k in range(reps):
        mpc.x0 = ex.X0; mpc.u0 = np.zeros(3); mpc._t0 = mpc._t0 * 0; mpc.set_initial_guess()
        t = time.perf_counter(); u0 = mpc.make_step(ex.X0); ts.append((time.perf_counter() - t) * 1e3)
    print("%-40s best %.2f ms (all %s) it=%d %s" % (label, min(ts), " ".join("%.1f" % t for t in ts), mpc.solver_stats["iter_count"], mpc.solver_stats["return_status"]), flush=True)
    return mpc.opt_x_num.master.copy()


out = {}
for defs in ("", "DOMPC_HIER_BARRIER=1"):
    if defs: os.environ["DOMPC_DEFS"] = defs
    else: os.environ.pop("DOMPC_DEFS", None)
    for kw, lab in ((dict(n_robust=5, uncertainty="paired"), "tree"), ({}, "9-scenario")):
        mpc = ex.build_mpc(ex.build_model(), **kw)
        for rep in range(2):
            x = run(mpc, "%s [%s]" % (lab, defs or "flat barrier"), reps=1 if rep == 0 else 5)
            if not mpc.solver_stats["success"]:
                break
        out.setdefault(lab, []).append(x)
        del mpc
for lab, xs in out.items():
    print(lab, "same bits:", bool(np.array_equal(xs[0], xs[1])))
