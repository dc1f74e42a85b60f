#!/usr/bin/env python3
"""MPC steps/s per BASELINE config and batch size, cold start (first step)
and warm start (4 subsequent steps) reported separately.  Kernel time = HIP events around the
batched device call with every input resident in HBM.

Warm steps close the loop on the MPC's own prediction (x0_next = predicted state at k=1 of
scenario 0, u_prev = applied u0, initial guess = previous solution, unshifted like
optimizer.py:754-768) - there is no plant integrator on the hot path.

usage: gpu_config_table.py [out.md] [Bmax]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from do_mpc_amd.examples import CASES  # noqa: E402
from do_mpc_amd.solver import STATS_DTYPE  # noqa: E402

CONFIGS = [
    ("configs[0] oscillating_masses N=10", "oscillating_masses", {}),
    ("configs[1] CSTR nominal deg 3", "CSTR", {"n_robust": 0, "collocation_deg": 3}),
    ("configs[2] batch_reactor N=50", "batch_reactor", {"n_horizon": 50}),
    ("configs[3] industrial_poly 9 scen. (A)", "industrial_poly", {}),
    ("configs[3] industrial_poly 9 scen. (B)", "industrial_poly", {"n_robust": 2, "uncertainty": "paired"}),
]


def x0_batch(name, ex, mpc, B):
    if name == "industrial_poly":
        return bench.synthetic_x0_batch(B)
    rng = np.random.default_rng(99)
    xi = rng.uniform(-1, 1, size=(B, ex.X0.size))
    X0 = ex.X0[None, :] * (1 + 0.02 * xi)
    return X0


def run(label, name, kw, Bs, out):
    import torch
    ex = CASES[name]
    mpc = ex.build_mpc(ex.build_model(), max_batch=max(Bs), **kw)
    ps, S = mpc.structure, mpc.S
    dev = torch.device("cuda", 0)
    xs = torch.from_numpy(mpc._x_scaling.master.copy()).to(dev)
    us = torch.from_numpy(mpc._u_scaling.master.copy()).to(dev)
    tlbx = torch.from_numpy(mpc._lb_opt_x.master).to(dev)
    tubx = torch.from_numpy(mpc._ub_opt_x.master).to(dev)
    tlbg = torch.from_numpy(mpc._nlp_cons_lb).to(dev)
    tubg = torch.from_numpy(mpc._nlp_cons_ub).to(dev)
    stream = torch.cuda.current_stream()
    for B in Bs:
        X0 = x0_batch(name, ex, mpc, B)
        P = np.tile(mpc.opt_p_num.master, (B, 1))
        P[:, :ps.nx] = X0
        P[:, ps.p_off_tvp:ps.p_off_p] = mpc.tvp_fun(0.0).master
        P[:, ps.p_off_p:ps.p_off_uprev] = mpc.p_fun(0.0).master
        Xi = np.zeros((B, ps.n_opt_x))
        Xi[:, :ps.off_z].reshape(B, -1, ps.nx)[:] = (X0 / mpc._x_scaling.master)[:, None, :]
        tXi = torch.from_numpy(Xi).to(dev)
        tP = torch.from_numpy(P).to(dev)
        tX = torch.empty((B, ps.n_opt_x), dtype=torch.float64, device=dev)
        tF = torch.empty(B, dtype=torch.float64, device=dev)
        tStats = torch.zeros(B * STATS_DTYPE.itemsize, dtype=torch.uint8, device=dev)

        def step(guess):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            S.solve_batch_device(B, guess.data_ptr(), tlbx.data_ptr(), tubx.data_ptr(), tlbg.data_ptr(),
                                 tubg.data_ptr(), tP.data_ptr(), tX.data_ptr(), 0, 0, 0, tF.data_ptr(),
                                 tStats.data_ptr(), stream=stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            st = np.frombuffer(tStats.cpu().numpy().tobytes(), dtype=STATS_DTYPE)
            step.sweeps = float(st["n_sweeps"].sum())
            return e0.elapsed_time(e1), int(st["success"].sum()), float(st["iter_count"].mean())

        step(tXi)                                   # untimed: code object load, clocks
        ms, ok, it = step(tXi)
        # roofline fraction of the cold launch (review r5: per config): algorithmic bytes of the model-evaluation sweeps (SURVEY 8(d) formula,
        # bench.sweep_bytes_per_problem) x the sweeps the launch executed / kernel time / 8 TB/s
        frac = bench.sweep_bytes_per_problem(ps) * step.sweeps / (ms * 1e-3) / 8e12
        row = {"config": label, "B": B, "cold_ms": ms, "cold_steps_s": B / ms * 1e3, "cold_ok": ok, "cold_iters": it, "cold_frac": frac}
        wms, wok, wit = [], [], []
        guess = torch.empty_like(tX)
        for _ in range(4):
            i1 = ps.ix(1, 0, ps.M)
            tP[:, :ps.nx] = tX[:, i1:i1 + ps.nx] * xs
            tP[:, ps.p_off_uprev:] = tX[:, ps.iu(0, 0):ps.iu(0, 0) + ps.nu] * us
            guess.copy_(tX)
            ms, ok, it = step(guess)
            wms.append(ms), wok.append(ok), wit.append(it)
        row.update({"warm_ms": float(np.mean(wms)), "warm_steps_s": B / float(np.mean(wms)) * 1e3,
                    "warm_ok": int(np.min(wok)), "warm_iters": float(np.mean(wit))})
        out.append(row)
        print(json.dumps(row), flush=True)
        del tXi, tP, tX, guess
        torch.cuda.empty_cache()


def run_mhe(Bs, out):
    """moving horizon estimation (do_mpc_amd.estimator.MHE, rotating masses, horizon 10, one estimated parameter): B estimation
    problems per launch - the five problems of the reference's stored run with perturbed measurement windows; "cold" = from the
    documented initial guess (x = 0, p_est = 1e-4), "warm" = from the previous stored solution like the reference's loop"""
    import torch
    ex = CASES["rotating_masses"]
    g = np.load(os.path.join(ROOT, "tests", "golden", "rotating_masses.npz"))
    OX, OP = g["estimator._opt_x_num"], g["estimator.opt_p_num"]
    mhe = ex.build_mhe(ex.build_model(), max_batch=max(Bs))
    mpc, S, ps = mhe._mpc, mhe.S, mhe._ps
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)      # noqa: E731
    tlbx, tubx, tlbg, tubg = t(mpc._lb_opt_x.master), t(mpc._ub_opt_x.master), t(mpc._nlp_cons_lb), t(mpc._nlp_cons_ub)
    stream = torch.cuda.current_stream()
    rng = np.random.default_rng(5)
    for B in Bs:
        idx = 1 + np.arange(B) % 4                                   # problems 1..4 (a stored previous solution exists)
        P_ref = OP[idx].copy()
        P_ref[:, mhe._po_y:] += 1e-3 * rng.standard_normal((B, P_ref.shape[1] - mhe._po_y))
        init0 = np.zeros(mhe.n_opt_x)
        init0[mhe._o_p:] = 1e-4
        tP = t(mhe._p_to_chain(P_ref))
        guesses = {"cold": t(mhe._to_chain(np.tile(init0, (B, 1)))), "warm": t(mhe._to_chain(OX[idx - 1]))}
        tX = torch.empty((B, ps.n_opt_x), dtype=torch.float64, device=dev)
        tF = torch.empty(B, dtype=torch.float64, device=dev)
        tStats = torch.zeros(B * STATS_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        row = {"config": "MHE rotating masses N=10, 1 estimated parameter", "B": B, "cold_frac": float("nan")}
        for kind, guess in guesses.items():
            for rep in range(2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                S.solve_batch_device(B, guess.data_ptr(), tlbx.data_ptr(), tubx.data_ptr(), tlbg.data_ptr(), tubg.data_ptr(),
                                     tP.data_ptr(), tX.data_ptr(), 0, 0, 0, tF.data_ptr(), tStats.data_ptr(), stream=stream.cuda_stream)
                e1.record(stream)
                torch.cuda.synchronize()
            st = np.frombuffer(tStats.cpu().numpy().tobytes(), dtype=STATS_DTYPE)
            ms = e0.elapsed_time(e1)
            row.update({kind + "_ms": ms, kind + "_steps_s": B / ms * 1e3, kind + "_ok": int(st["success"].sum()), kind + "_iters": float(st["iter_count"].mean())})
        out.append(row)
        print(json.dumps(row), flush=True)


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else None
    bmax = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    Bs = [b for b in (1, 64, 1024, 4096, 16384) if b <= bmax]
    rows = []
    only = os.environ.get("DOMPC_TABLE_ONLY", "")                  # "mhe": the estimator rows only
    for label, name, kw in CONFIGS:
        if only and only not in label: continue
        run(label, name, kw, Bs, rows)
    run_mhe([b for b in Bs if b <= 4096], rows)
    lines = ["| config | B | cold ms | cold steps/s | cold conv. | cold iters | cold roofline frac | warm ms | warm steps/s | warm conv. | warm iters |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append(f"| {r['config']} | {r['B']} | {r['cold_ms']:.2f} | {r['cold_steps_s']:.0f} | {r['cold_ok']}/{r['B']} | "
                     f"{r['cold_iters']:.1f} | {('%.4f' % r['cold_frac']) if r['cold_frac'] == r['cold_frac'] else '-'} | {r['warm_ms']:.2f} | {r['warm_steps_s']:.0f} | {r['warm_ok']}/{r['B']} | {r['warm_iters']:.1f} |")
    txt = "\n".join(lines) + "\n"
    print(txt)
    if path:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as fh:
            fh.write(txt)


if __name__ == "__main__":
    main()
