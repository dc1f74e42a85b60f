"""Members of the timed batch (B = 16384) against oracle solves, for a list of DOMPC_DEFS build variants: iteration counts per member,
batch mean, kernel time.   python tools/gpu_timed_members.py "" "DOMPC_ADJ_REFINE=1" ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import multiprocessing as mp
import numpy as np


def _oracle_member(args):
    import warnings
    warnings.filterwarnings("ignore")
    import parity_common as pc
    from oracle import ipm as oipm
    i, x0 = args
    nlp = pc.oracle_nlp("industrial_poly")
    r = oipm.solve(nlp, nlp.initial_guess(x0), nlp.opt_p(x0, np.zeros(nlp.nu)))
    return i, nlp.u0_of(r["x"]), int(r["stats"]["iter_count"]), bool(r["stats"]["success"]), r["x"]


if __name__ == "__main__":
    import bench
    import parity_common as pc
    from do_mpc_amd.examples import industrial_poly as ex
    B = 16384
    X0 = bench.synthetic_x0_batch(B)
    members = [0, 1, 2047, 2048, 4095, 8191, 12345, 16383, 100, 5000, 9999, 15000]
    with mp.get_context("spawn").Pool(len(members)) as pool:
        res = pool.map(_oracle_member, [(i, X0[i]) for i in members])
    print("oracle iterations", {i: it for i, _, it, _, _ in res}, flush=True)
    for defs in sys.argv[1:] or [""]:
        os.environ["DOMPC_DEFS"] = defs
        mpc = ex.build_mpc(ex.build_model(), max_batch=B)
        used = np.ones(mpc.structure.n_opt_x, bool)
        used[mpc.structure.tables["dummy_idx"]] = False
        r = mpc.make_step_batch(X0)
        t0 = time.perf_counter()
        r = mpc.make_step_batch(X0)
        dt = time.perf_counter() - t0
        it = r["stats"]["iter_count"]
        print("== DOMPC_DEFS=%r: success %d/%d, iterations mean %.3f max %d, %.1f ms, inf_du max %.2e" %
              (defs, int(r["stats"]["success"].sum()), B, it.mean(), it.max(), dt * 1e3,
               float(np.max(r["stats"]["inf_du"])) if "inf_du" in r["stats"].dtype.names else -1.0), flush=True)
        for i, u_ref, it_ref, ok, x_ref in res:
            print("   member %5d: iterations %d (oracle %d)  u0 err %.1e  x err %.1e" %
                  (i, it[i], it_ref, pc.relerr(r["u0"][i], u_ref), pc.relerr(r["x"][i][used], x_ref[used])), flush=True)
        del mpc
