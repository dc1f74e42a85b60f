"""Randomly drawn members of the timed batch (B = 16384, bench.synthetic_x0_batch) against oracle solves of the same x0: iteration counts,
u0 and the full primal solution.   python tools/gpu_random_members.py [n_members] [seed] [DOMPC_DEFS]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import multiprocessing as mp
import numpy as np
from gpu_timed_members import _oracle_member

if __name__ == "__main__":
    import bench
    import parity_common as pc
    from do_mpc_amd.examples import industrial_poly as ex
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    if len(sys.argv) > 3 and sys.argv[3]:
        os.environ["DOMPC_DEFS"] = sys.argv[3]
    B = 16384
    X0 = bench.synthetic_x0_batch(B)
    members = sorted(int(i) for i in np.random.default_rng(seed).choice(B, size=n, replace=False))
    if os.environ.get("DOMPC_MEMBERS"):                    # named members instead of drawn ones: DOMPC_MEMBERS=3526,7283
        members = sorted(int(i) for i in os.environ["DOMPC_MEMBERS"].split(","))
        n = len(members)
    mpc = ex.build_mpc(ex.build_model(), max_batch=B)
    used = np.ones(mpc.structure.n_opt_x, bool)
    used[mpc.structure.tables["dummy_idx"]] = False
    r = mpc.make_step_batch(X0)
    it = r["stats"]["iter_count"]
    print("batch: success %d/%d, iterations mean %.4f max %d" % (int(r["stats"]["success"].sum()), B, it.mean(), it.max()), flush=True)
    t0 = time.time()
    with mp.get_context("spawn").Pool(min(n, os.cpu_count() or 8)) as pool:
        res = pool.map(_oracle_member, [(i, X0[i]) for i in members])
    print("oracle solves: %.0f s on %d processes" % (time.time() - t0, min(n, os.cpu_count() or 8)), flush=True)
    late = []
    worst_x = 0.0
    for i, u_ref, it_ref, ok, x_ref in res:
        ex_ = pc.relerr(r["x"][i][used], x_ref[used])
        worst_x = max(worst_x, ex_)
        if it[i] != it_ref:
            late.append((i, int(it[i]), it_ref))
        print("   member %5d: iterations %d (oracle %d)%s  u0 err %.1e  x err %.1e" %
              (i, it[i], it_ref, "" if it[i] == it_ref else "  <<<", pc.relerr(r["u0"][i], u_ref), ex_), flush=True)
    print("members with another iteration count: %d of %d %s; worst x err %.1e" % (len(late), n, late, worst_x))
