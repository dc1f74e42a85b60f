#!/usr/bin/env python3
"""Tree-sharded solve on ONE GPU (world = 1): exercises the device<->host exchange handshake (pinned-memory
request/acknowledge words, host service loop) and the cut-parent code path against the unsharded solve.
 (a) exchange = identity callback, (b) exchange = torch.distributed all_reduce on an nccl (RCCL) group of size 1."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from do_mpc_amd.examples import CASES  # noqa: E402


def solve(name, kw, shard=None):
    ex = CASES[name]
    mpc = ex.build_mpc(ex.build_model(), **kw)
    mpc.x0 = ex.X0
    mpc.set_initial_guess()
    if shard:
        mpc.shard_tree(**shard)
    t = time.time()
    u0 = mpc.make_step(ex.X0).ravel().copy()
    dt = time.time() - t
    t = time.time()
    mpc.make_step(ex.X0)
    dt2 = time.time() - t
    return u0, mpc.opt_x_num.master.copy(), dict(mpc.solver_stats), dt, dt2, mpc.structure.tables["dummy_idx"]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "industrial_poly"
    kw = {"n_robust": 2, "uncertainty": "paired"} if name == "industrial_poly" else {}
    cut = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    u_ref, x_ref, st_ref, dt, dt2, dummy = solve(name, kw)
    keep = np.ones(x_ref.size, bool)
    keep[dummy] = False
    print(f"unsharded      : u0={u_ref} it={st_ref['iter_count']} {st_ref['return_status']} {dt * 1e3:.1f} ms / warm {dt2 * 1e3:.1f} ms", flush=True)
    calls = []
    u, x, st, dt, dt2, _ = solve(name, kw, dict(rank=0, world=1, cut_level=cut, allreduce=lambda v: calls.append(v.numel())))
    print(f"cut, identity  : u0={u} it={st['iter_count']} {st['return_status']} {dt * 1e3:.1f} ms / warm {dt2 * 1e3:.1f} ms, "
          f"{len(calls)} exchanges, max|dx|={np.abs(x - x_ref)[keep].max():.2e}", flush=True)
    assert st["success"] and np.allclose(u, u_ref, rtol=1e-8)
    u, x, st, dt, dt2, _ = solve(name, kw, dict(rank=0, world=1, cut_level=cut))            # native RCCL (default)
    print(f"cut, native RCCL: u0={u} it={st['iter_count']} {st['return_status']} {dt * 1e3:.1f} ms / warm {dt2 * 1e3:.1f} ms, "
          f"max|dx|={np.abs(x - x_ref)[keep].max():.2e}", flush=True)
    assert st["success"] and np.allclose(u, u_ref, rtol=1e-8)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    u, x, st, dt, dt2, _ = solve(name, kw, dict(rank=0, world=1, cut_level=cut, native_rccl=False))
    print(f"cut, torch nccl: u0={u} it={st['iter_count']} {st['return_status']} {dt * 1e3:.1f} ms / warm {dt2 * 1e3:.1f} ms, "
          f"max|dx|={np.abs(x - x_ref)[keep].max():.2e}", flush=True)
    assert st["success"] and np.allclose(u, u_ref, rtol=1e-8)
    dist.destroy_process_group()
    print("OK")


if __name__ == "__main__":
    main()
