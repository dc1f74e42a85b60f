"""Solve times of the dense edge path (DAE models, estimator): cold make_step at B = 1 and a batch, iterations and first input.
Usage: python tools/gpu_dense_time.py [B]      (prebuilt code objects of __graft_entry__.PREBUILT / PREBUILT_MHE)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from do_mpc_amd.examples import CASES

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def run(name, kw, builder="build_mpc", mkw=None):
    ex = CASES[name]
    obj = getattr(ex, builder)(ex.build_model(**(mkw or {})), max_batch=B, **kw)
    mpc = getattr(obj, "_mpc", obj)
    rng = np.random.default_rng(0)
    if builder == "build_mpc":
        x0 = np.asarray(ex.X0, float).ravel()
        for Bq in (1, B):
            X0 = np.tile(x0, (Bq, 1)) * (1.0 + 1e-3 * rng.standard_normal((Bq, x0.size)))
            best = 1e9
            for rep in range(3):
                t = time.perf_counter()
                r = mpc.make_step_batch(X0)
                best = min(best, (time.perf_counter() - t) * 1e3)
            st = r["stats"]
            print(f"{name} {kw} B={Bq}: {best:.2f} ms  converged {int(np.sum(st['success']))}/{Bq}  iters {np.mean(st['iter_count']):.1f}  u0[0] {r['u0'][0]!r}", flush=True)
    else:
        mhe = obj
        g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "rotating_masses.npz"))
        P = np.asarray(g["estimator.opt_p_num"], float)[3][None, :]
        init = np.asarray(g["estimator._opt_x_num"], float)[2][None, :]
        for Bq in (1, B):
            Pq = np.tile(P, (Bq, 1))
            Pq[:, mhe._po_y:] += 1e-3 * rng.standard_normal((Bq, Pq.shape[1] - mhe._po_y))
            best = 1e9
            for rep in range(3):
                t = time.perf_counter()
                r = mhe.solve_batch(Pq, np.tile(init, (Bq, 1)))
                best = min(best, (time.perf_counter() - t) * 1e3)
            st = r["stats"]
            print(f"MHE {name} B={Bq}: {best:.2f} ms  converged {int(np.sum(st['success']))}/{Bq}  iters {np.mean(st['iter_count']):.1f}", flush=True)


run("dip", {})
run("oscillating_masses_dae", {})
run("CSTR", {"n_robust": 0, "nl_cons_check_colloc_points": True})
run("rotating_masses", {}, builder="build_mhe")
