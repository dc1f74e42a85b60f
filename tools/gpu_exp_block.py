#!/usr/bin/env python3
"""Batch throughput of the cold industrial_poly workload vs threads per problem (block_threads) and batch size.
usage: gpu_exp_block.py [blocks=256,128,64] [batches=1024,4096,8192] [steps=2]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from do_mpc_amd.examples import industrial_poly as ex  # noqa: E402
from do_mpc_amd.solver import STATS_DTYPE  # noqa: E402

blocks = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256,128,64").split(",")]
batches = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1024,4096,8192").split(",")]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda", 0)
for block in blocks:
    for B in batches:
        mpc = ex.build_mpc(ex.build_model(), max_batch=B, block_threads=block)
        ps, S = mpc.structure, mpc.S
        X0 = bench.synthetic_x0_batch(B)
        P = np.tile(mpc.opt_p_num.master, (B, 1))
        P[:, :ps.nx] = X0
        P[:, ps.p_off_p:ps.p_off_uprev] = mpc.p_fun(0.0).master
        Xi = np.zeros((B, ps.n_opt_x))
        Xi[:, :ps.off_z].reshape(B, -1, ps.nx)[:] = (X0 / mpc._x_scaling.master)[:, None, :]
        t = {k: torch.from_numpy(v).to(dev) for k, v in dict(x0=Xi, p=P, lbx=mpc._lb_opt_x.master, ubx=mpc._ub_opt_x.master,
                                                             lbg=mpc._nlp_cons_lb, ubg=mpc._nlp_cons_ub).items()}
        tX = torch.empty((B, ps.n_opt_x), dtype=torch.float64, device=dev)
        tF = torch.empty(B, dtype=torch.float64, device=dev)
        tStats = torch.zeros(B * STATS_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream()

        def step():
            S.solve_batch_device(B, t["x0"].data_ptr(), t["lbx"].data_ptr(), t["ubx"].data_ptr(), t["lbg"].data_ptr(),
                                 t["ubg"].data_ptr(), t["p"].data_ptr(), tX.data_ptr(), 0, 0, 0, tF.data_ptr(),
                                 tStats.data_ptr(), stream=stream.cuda_stream)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        st = np.frombuffer(tStats.cpu().numpy().tobytes(), dtype=STATS_DTYPE)
        print(f"block {block:4d} B {B:6d} slots {S.num_slots:5d}: {dt * 1e3:9.1f} ms/step  {B / dt:9.1f} steps/s  "
              f"ok {int(st['success'].sum())}/{B} iters {st['iter_count'].mean():.1f}", flush=True)
        del mpc, S, t, tX, tF, tStats
        torch.cuda.empty_cache()
