#!/usr/bin/env python3
"""Quick on-GPU sanity run: golden replay for every case + a small batch timing."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from do_mpc_amd.examples import BASELINE_CASES, CASES  # noqa: E402


def main():
    names = sys.argv[1:] or list(BASELINE_CASES)      # (the cases with golden vectors)
    for name in names:
        ex = CASES[name]
        t = time.time()
        model = ex.build_model()
        mpc = ex.build_mpc(model, max_batch=256)
        print(f"{name}: setup {time.time() - t:.1f}s n_opt_x={mpc.structure.n_opt_x} slots={mpc.S.num_slots} "
              f"ws={mpc.S.workspace_bytes / 1e6:.1f}MB", flush=True)
        mpc.x0 = ex.X0
        mpc.set_initial_guess()
        g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        U, Xs = g["mpc._u"], g["mpc._x"]
        for k in range(5):
            t = time.time()
            u0 = mpc.make_step(Xs[k])
            st = mpc.solver_stats
            err = np.max(np.abs(u0.ravel() - U[k]) / np.maximum(1, np.abs(U[k])))
            print(f"  step {k}: {st['return_status']} it={st['iter_count']} reg={st['n_reg']} lsf={st['n_ls_fail']} "
                  f"t={time.time() - t:.4f}s relerr={err:.2e}", flush=True)
            mpc.u0 = U[k]
        rng = np.random.default_rng(99)
        for B in (1, 64, 256):
            X0 = ex.X0 * (1 + 0.02 * rng.uniform(-1, 1, size=(B, len(ex.X0))))
            t = time.time()
            r = mpc.make_step_batch(X0)
            dt = time.time() - t
            st = r["stats"]
            print(f"  batch {B}: {dt:.3f}s  {B / dt:.1f} solves/s  success={st['success'].mean():.2f} "
                  f"iters mean={st['iter_count'].mean():.1f} max={st['iter_count'].max()}", flush=True)


if __name__ == "__main__":
    main()
