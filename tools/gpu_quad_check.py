#!/usr/bin/env python3
"""GPU check of the four-edges-per-wavefront sweep (csrc/dompc_quad.h) against the wavefront-per-edge path it replaces
(the same kernels built with -DDOMPC_QUAD=0): sweep outputs (g, [A B | c | Q~] blocks), one Newton direction, a cold solve.
   python tools/gpu_quad_check.py [case] [B]          (case: a name of do_mpc_amd.examples.CASES, default industrial_poly)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def build(ex, quad, max_batch):
    if quad:
        os.environ.pop("DOMPC_DEFS", None)
    else:
        os.environ["DOMPC_DEFS"] = "DOMPC_QUAD=0"
    mpc = ex.build_mpc(ex.build_model(), max_batch=max_batch)
    os.environ.pop("DOMPC_DEFS", None)
    return mpc


def main():
    import torch
    from do_mpc_amd.examples import CASES
    name = sys.argv[1] if len(sys.argv) > 1 else "industrial_poly"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    ex = CASES[name]
    dev = torch.device("cuda", 0)
    out = {}
    rng = np.random.default_rng(7)
    for quad in (1, 0):
        mpc = build(ex, quad, max(B, 64))
        ps, S = mpc.structure, mpc.S
        mpc.x0 = ex.X0
        mpc.set_initial_guess()
        if quad:
            # iterates strictly inside the bounds, multipliers of order one
            x_ref = np.zeros(ps.n_opt_x)
            x_ref[:ps.off_z].reshape(-1, ps.nx)[:] = ex.X0 / np.asarray(mpc._x_scaling.master)
            x_ref[ps.off_u:ps.off_eps].reshape(-1, ps.nu)[:] = 0.5 * (np.clip(np.asarray(mpc._u_lb.master), -1e3, 1e3) + np.clip(np.asarray(mpc._u_ub.master), -1e3, 1e3))
            X = np.tile(x_ref, (B, 1)) * (1 + 0.01 * rng.uniform(-1, 1, size=(B, ps.n_opt_x)))
            L = rng.normal(size=(B, ps.n_g))
            P = np.tile(np.asarray(mpc.opt_p_num.master).ravel(), (B, 1))
            P[:, :ps.nx] = ex.X0
            if hasattr(mpc, "p_fun") and ps.p_off_uprev > ps.p_off_p:
                P[:, ps.p_off_p:ps.p_off_uprev] = np.asarray(mpc.p_fun(0.0).master).ravel()
            out["X"], out["L"], out["P"] = X, L, P
        X, L, P = out["X"], out["L"], out["P"]
        tX, tL, tP = (torch.from_numpy(a).to(dev) for a in (X, L, P))
        nb = S.sweep_block_doubles
        tG = torch.zeros((B, ps.n_g), dtype=torch.float64, device=dev)
        tB = torch.zeros((B, ps.n_edges, nb), dtype=torch.float64, device=dev)
        S.sweep_batch_device(B, tX.data_ptr(), tL.data_ptr(), tP.data_ptr(), tG.data_ptr(), tB.data_ptr())
        torch.cuda.synchronize()
        out[("g", quad)] = tG.cpu().numpy()
        out[("blocks", quad)] = tB.cpu().numpy()
        # one Newton direction at the first iterate
        lbx, ubx = np.asarray(mpc._lb_opt_x.master).ravel(), np.asarray(mpc._ub_opt_x.master).ravel()
        lo = np.where(np.isfinite(lbx), lbx, -1e20)
        hi = np.where(np.isfinite(ubx), ubx, 1e20)
        x1 = np.clip(X[0], lo + 1e-3 * np.maximum(1, np.abs(lo)), hi - 1e-3 * np.maximum(1, np.abs(hi)))
        zl, zu = np.ones(ps.n_opt_x), np.ones(ps.n_opt_x)
        try:
            dx, dl, rd, c = S.debug_newton_step(x1, L[0], zl, zu, lbx, ubx, np.zeros(ps.n_g), np.zeros(ps.n_g), P[0], 0.1, 0.0)
            out[("dx", quad)], out[("dl", quad)], out[("rd", quad)] = dx, dl, rd
        except Exception as exc:                                   # noqa: BLE001
            print("debug_newton_step:", exc)
        t = time.time()
        u0 = mpc.make_step(ex.X0)
        st = mpc.solver_stats
        print(f"quad={quad}: {st['return_status']} it={st['iter_count']} reg={st.get('n_reg')} t={time.time() - t:.3f}s u0={np.asarray(u0).ravel()}", flush=True)
        out[("u0", quad)] = np.asarray(u0).ravel().copy()
        out[("x", quad)] = np.asarray(mpc.opt_x_num.master).ravel().copy() if hasattr(mpc.opt_x_num, "master") else None
        Xb = ex.X0 * (1 + 0.01 * np.random.default_rng(3).uniform(-1, 1, size=(B, len(ex.X0))))
        r = mpc.make_step_batch(Xb)
        out[("ub", quad)] = np.asarray(r["u0"]).copy()
        out[("itb", quad)] = r["stats"]["iter_count"].copy()
        print(f"   batch: success {r['stats']['success'].mean():.2f} iters {r['stats']['iter_count']}", flush=True)
        del mpc, S

    def rel(a, b):
        return float(np.max(np.abs(a - b) / (1e-300 + np.maximum(1.0, np.maximum(np.abs(a), np.abs(b))))))
    print("g       :", rel(out[("g", 1)], out[("g", 0)]))
    bq, bo = out[("blocks", 1)], out[("blocks", 0)]
    print("blocks  :", rel(bq, bo), " worst entry (b, e, i):", np.unravel_index(np.argmax(np.abs(bq - bo) / np.maximum(1, np.abs(bo))), bq.shape))
    for k in ("dx", "dl", "rd"):
        if (k, 1) in out and (k, 0) in out:
            print(f"{k:8s}:", rel(out[(k, 1)], out[(k, 0)]))
    print("u0      :", rel(out[("u0", 1)], out[("u0", 0)]))
    print("u0 batch:", rel(out[("ub", 1)], out[("ub", 0)]), " iterations equal:", bool(np.array_equal(out[("itb", 1)], out[("itb", 0)])))


if __name__ == "__main__":
    main()
