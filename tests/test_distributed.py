"""N>1 path of bench.py on CPU: world_size-2 gloo group, batch sharding and the max-over-ranks
timing reduction (the only collectives on this path; the x0 batch shards with no data exchange)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench


def test_shards_partition_the_batch_exactly():
    for B, W in [(10, 3), (512, 8), (7, 8), (1024, 2)]:
        spans = [bench.shard(B, r, W) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == B
        assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_synthetic_batch_is_rank_independent_and_in_bounds():
    X = bench.synthetic_x0_batch(64)
    assert np.array_equal(X, bench.synthetic_x0_batch(64))
    assert np.all(np.abs(X[:, 3] - 363.15) <= 1.0 + 1e-9)           # stays inside the +-2 K band
    lo, hi = bench.shard(64, 1, 2)
    assert np.array_equal(bench.synthetic_x0_batch(64)[lo:hi], X[32:])


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = bench.shard(20, rank, world)
    X = bench.synthetic_x0_batch(20)[lo:hi]
    t = torch.tensor([1.0 + rank], dtype=torch.float64)         # pretend rank r needed (1+r) s
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    cnt = torch.tensor([float(X.shape[0])], dtype=torch.float64)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    q.put((rank, float(t.item()), float(cnt.item()), float(X.sum())))
    dist.destroy_process_group()


def test_two_rank_gloo_timing_reduction_and_coverage():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 2.0                      # max over ranks
    assert res[0][2] == res[1][2] == 20.0                     # every problem owned by exactly one rank
    assert abs(res[0][3] + res[1][3] - bench.synthetic_x0_batch(20).sum()) < 1e-6


def _run_bench(extra_env, *argv, timeout=900):
    import subprocess
    import sys
    env = dict(os.environ, **extra_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "bench.py"), *argv],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_gpus_flag_launches_that_many_ranks_and_reports_them():
    """`python bench.py --gpus 2` started plainly (the driver's N = 1 command line with another N): the script launches two
    ranks of itself - here on the host emulation of the kernels with a gloo group (DOMPC_BENCH_BACKEND=hostemu, test plumbing) -
    and rank 0 prints ONE JSON line with n_gpus = 2, the whole-job batch and a partition of it (the flag was parsed
    and dropped)."""
    import json
    r = _run_bench({"DOMPC_BENCH_BACKEND": "hostemu"}, "--gpus", "2", "--batch", "2", "--steps", "1", "--warmup", "0",
                   "--no-cpu-baseline", "--no-b1", "--no-variant-b", "--no-traffic", "--sweep-steps", "1", "--sweep-warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["metric"].startswith("MPC steps/sec")
    assert out["config"]["global_batch"] == 4 and out["config"]["batch_per_gpu"] == 2
    assert out["config"]["shard_of_rank0"] == [0, 2]
    assert out["solve"]["converged_all_ranks"] == 4 and out["solve"]["of_all_ranks"] == 4
    assert abs(out["value"] - 4 * out["steps"] / (out["ms_per_step"] * out["steps"] * 1e-3)) < 1e-6 * out["value"]
    assert "HOST EMULATION" in out["data"]
    assert out["roofline"]["sweep_only"]["achieved"] > 0


def test_gpus_flag_fails_loudly_without_the_devices():
    """the product path: no time-sharing of one GPU between ranks, no silent n_gpus = 1"""
    r = _run_bench({"DOMPC_BENCH_BACKEND": "hip"}, "--gpus", "2", "--steps", "1", "--warmup", "0", timeout=300)
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two devices are visible")
    assert r.returncode != 0
    assert "needs 2 devices" in (r.stderr + r.stdout)
