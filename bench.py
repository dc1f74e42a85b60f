#!/usr/bin/env python3
"""bench.py - MPC steps/sec of the make_step hot path on MI355X.

Workload (BASELINE.json configs[3]): industrial_poly robust multi-stage NMPC, 9 scenarios
(shipped tree: 9 parameter combinations x n_robust=1, N=20, Radau deg 2; the golden-pinned
variant).  One "step" = one batched make_step: B independent problems (synthetic x0 batch,
cold-started from MPC.set_initial_guess semantics) solved by one launch of the persistent IPM
kernel with every input already resident in HBM.  value = B * steps * n_gpus / time.

Multi-GPU: the x0 batch shards across ranks with no data-path collective (weak scaling:
B problems per GPU); the only collectives are the timing barrier and the max-over-ranks reduce.

`--variant tree`: ONE industrial_poly problem with a 3^n_robust-leaf scenario tree (default n_robust=5: the
243-leaf tree of BASELINE.json configs[4]) whose sub-trees are sharded over the ranks (strong scaling): the ranks
exchange cut-edge contributions of the Riccati recursion and the scalar reductions of the IPM through RCCL
all-reduces (torch.distributed, backend nccl) - SURVEY.md 8(e).  value = MPC steps/s of that single problem.

Extra fields: `roofline` for the dominant kernel (dompc_solve_kernel; HBM-bound model:
algorithmic bytes of the derivative sweeps it executed / its HIP-event duration) and
`cpu_baseline` (the CPU oracle, a port of the reference algorithm, timed on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

# TEST-ONLY switch (tests/test_distributed.py): DOMPC_BENCH_BACKEND=hostemu runs the same script on the g++ host emulation of
# the kernels (tests/hostemu.py) with gloo instead of RCCL, so that the N-rank launch path, the batch partition and the one
# JSON line can be checked in the GPU-less CI container.  Such a line says so in `data`; it is not a measurement.
BACKEND = os.environ.get("DOMPC_BENCH_BACKEND", "hip")


def synthetic_x0_batch(B: int, seed: int = 99) -> np.ndarray:
    """x0_i around the example's initial state (seed 99): masses x (1 + 2% U(-1,1)),
    temperatures +- 1 K U(-1,1) (a 2 % *relative* perturbation of Kelvin temperatures leaves the
    +-2 K reactor band and makes the robust problem infeasible), T_adiab recomputed."""
    from do_mpc_amd.examples import industrial_poly as ex
    rng = np.random.default_rng(seed)
    X0 = np.tile(ex.X0, (B, 1))
    xi = rng.uniform(-1, 1, size=(B, 10))
    X0[:, [0, 1, 2, 8]] *= (1 + 0.02 * xi[:, [0, 1, 2, 8]])
    X0[:, 3:8] += xi[:, 3:8]
    X0[:, 9] = X0[:, 1] * 950.0 / ((X0[:, 0] + X0[:, 1] + X0[:, 2]) * 5.0) + X0[:, 3]
    return X0


def shard(B_total: int, rank: int, world: int):
    """contiguous shard [lo, hi) of a global batch"""
    per = B_total // world
    rem = B_total % world
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def sweep_bytes_per_problem(ps) -> int:
    """Algorithmic bytes of one model-evaluation sweep of one problem (SURVEY.md 8(d) formula)."""
    nx, nu, nz, np_, ntvp, ne = ps.nx, ps.nu, 0, ps.np_, ps.ntvp, ps.ne
    M = ps.M
    d = ps.ni * ps.deg if M else 1
    n_v = nx + nz + nu
    T = lambda n: n * (n + 1) // 2  # noqa: E731
    n_eps_e = ps.ns
    reads = (M + 2) * nx + M * nz + nu + n_eps_e + (M * (nx + nz) + nx + ne) + (np_ + ntvp)
    writes = (M * (nx + nz) + nx + ne) + d * (nx + nz) * n_v + d * T(n_v) + T(nx + nu + n_eps_e) + (nx + nu + n_eps_e)
    return 8 * (reads + writes) * ps.n_edges


# ---- CPU baselines (rank 0, N = 1 only; bounded samples).  One spawned process builds the problem once (no fork after the HIP
#      runtime is up in the parent) and then FORKS one single-threaded worker per host core - the way the reference itself
#      parallelises make_step (examples/tools/sampling/multiprocessing/closed_loop/mp_sampling_closed_loop_02.py:69-70).
_CPU_STATE = {}


def _cpu_oracle_worker(job):
    """one process = one core: cold make_step solves of the oracle (scipy SuperLU and BLAS pinned to one thread)"""
    idx, n_per = job
    from oracle import ipm
    nlp, X0 = _CPU_STATE["nlp"], _CPU_STATE["X0"]
    out = []
    for i in range(idx * n_per, idx * n_per + n_per):
        t0 = time.perf_counter()
        r = ipm.solve(nlp, nlp.initial_guess(X0[i]), nlp.opt_p(X0[i], np.zeros(nlp.nu)), opts={"fast": True})
        out.append((int(r["stats"]["success"]), int(r["stats"]["iter_count"]), time.perf_counter() - t0))
    return out


def _cpu_hostemu_worker(job):
    """one process = one core: the product's structured algorithm compiled for the host (tests/_hostemu, g++ -O2), one thread"""
    idx, n_per = job
    mpc, X0 = _CPU_STATE["mpc"], _CPU_STATE["X0"]
    t0 = time.perf_counter()
    r = mpc.make_step_batch(X0[idx * n_per: idx * n_per + n_per])
    dt = time.perf_counter() - t0
    return [(int(ok), int(it), dt / n_per) for ok, it in zip(r["stats"]["success"], r["stats"]["iter_count"])]


def _cpu_fanout(kind, cores, per_core, q):
    """runs in a spawned process: build once, fork `cores` workers, collect (ok, iterations, seconds) per solve"""
    import multiprocessing as mp
    import warnings
    warnings.filterwarnings("ignore")
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"
    try:
        try:
            from threadpoolctl import threadpool_limits
            threadpool_limits(1)
        except Exception:       # noqa: BLE001
            pass
        _CPU_STATE["X0"] = synthetic_x0_batch(cores * per_core)
        t0 = time.perf_counter()
        if kind == "oracle":
            from oracle.models import CASES
            from oracle.nlp import OracleNLP
            _CPU_STATE["nlp"] = OracleNLP(CASES["industrial_poly"]())
            worker = _cpu_oracle_worker
        else:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import hostemu                               # TEST-ONLY build of the kernel text for the host (never the product path)
            from do_mpc_amd.examples import industrial_poly as ex
            with hostemu.patched():
                _CPU_STATE["mpc"] = ex.build_mpc(ex.build_model(), max_batch=per_core)
            worker = _cpu_hostemu_worker
        t_build = time.perf_counter() - t0
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(cores) as pool:
            res = pool.map(worker, [(i, per_core) for i in range(cores)], chunksize=1)
        q.put(("ok", res, t_build, time.perf_counter() - t0))
    except Exception as e:      # noqa: BLE001
        q.put(("error", repr(e), 0.0, 0.0))


def _run_fanout(kind, cores, per_core, timeout):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")                 # (no fork after the HIP runtime is up in THIS process)
    q = ctx.Queue()
    pr = ctx.Process(target=_cpu_fanout, args=(kind, cores, per_core, q))
    # one BLAS / OpenMP thread per process: the variables must be in the environment BEFORE the child imports numpy (a spawned
    # child imports this module first) - with the default thread pools 256 workers x 256 threads spin on each other
    # ... and glibc malloc / numpy told to keep freed memory and to leave transparent huge pages alone (fresh-page faults of the
    # numpy temporaries and of SuperLU's work arrays serialise in the kernel when every core runs a worker)
    child_env = {"OMP_NUM_THREADS": "1", "OPENBLAS_NUM_THREADS": "1", "MKL_NUM_THREADS": "1", "MALLOC_MMAP_THRESHOLD_": "33554432",
                 "MALLOC_TRIM_THRESHOLD_": "4294967296", "MALLOC_TOP_PAD_": "268435456", "MALLOC_ARENA_MAX": "1", "NUMPY_MADVISE_HUGEPAGE": "0"}
    saved = {v: os.environ.get(v) for v in child_env}
    os.environ.update(child_env)
    try:
        pr.start()
    finally:
        for v, old in saved.items():
            if old is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = old
    t_end = time.perf_counter() + timeout
    got = None
    while got is None:
        try:
            got = q.get(timeout=1.0)
        except Exception:       # noqa: BLE001  (queue.Empty)
            if not pr.is_alive():
                return None, "the fan-out process died (exit code %r)" % pr.exitcode
            if time.perf_counter() > t_end:
                pr.terminate()
                return None, "timed out after %.0f s" % timeout
    tag, res, t_build, t_par = got
    pr.join(timeout=30)
    if tag != "ok":
        return None, res
    flat = [x for r in res for x in r]
    n = len(flat)
    n_ok = sum(x[0] for x in flat)
    t_solve = sum(x[2] for x in flat)             # core-seconds inside the solves
    iters = sum(x[1] for x in flat)
    return {"n": n, "n_ok": n_ok, "per_core": n / t_solve, "value": n / t_par, "ms_per_iteration": 1e3 * t_solve / max(iters, 1),
            "s_per_solve": t_solve / n, "iters_mean": iters / n, "t_par": t_par, "t_build": t_build}, None


def cpu_baseline(per_core: int = 1, max_cores: int = 0) -> dict:
    """CPU side of the comparison, on ALL host cores (one single-threaded process per core), bounded samples of the same workload:
    (1) `value`: the CPU oracle (oracle/: restatement of the reference's NLP + IPOPT's algorithm on the FLAT sparse NLP, general
        sparse LU of the KKT matrix like IPOPT / MUMPS; kind "port") with its cheaper linear algebra switched on (oracle/ipm.py
        _FastKKT: structural singularity test instead of two failing factorisations per iteration, one RCM ordering per solve);
    (2) `same_algorithm_on_cpu`: the product's own structured algorithm (per-edge condensing + tree Riccati) compiled for the
        host by g++ -O2 (the TEST-ONLY host emulation of the kernel text, tests/_hostemu) - not the reference's path, reported
        because it is the stronger CPU number: what these cores do with the algorithm the GPU runs."""
    try:
        logical = len(os.sched_getaffinity(0))
    except AttributeError:
        logical = os.cpu_count() or 1
    # CPU time this process tree may actually use: the cgroup quota (cpu.max = "quota period"; the GPU boxes of the pool show 256
    # logical CPUs under a quota of 16 - more runnable workers than that are throttled: measured 78 tasks/s with 16 busy
    # processes, 57 with 256)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:           # noqa: BLE001
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:       # noqa: BLE001
            pass
    cores = logical if quota is None else max(1, min(logical, int(quota + 0.5)))
    host_cores = cores
    cap = max_cores or int(os.environ.get("DOMPC_CPU_BASELINE_CORES", "0"))
    if cap > 0:
        cores = min(cores, cap)
    t0 = time.perf_counter()
    per_core = max(per_core, 3 if cores <= 32 else 1)
    o, err = _run_fanout("oracle", cores, per_core, timeout=240.0)
    out = {"value": None, "unit": "MPC steps/s", "cores": cores, "host_cores": host_cores, "kind": "port",
           "host_logical_cpus": logical, "cpu_quota": quota}
    if o is None:
        out["sample"] = f"oracle fan-out failed: {err}"
    else:
        out.update({"value": o["value"], "per_core": o["per_core"], "ms_per_iteration": o["ms_per_iteration"],
                    "sample": f"{o['n']} cold make_step solves of the same workload ({per_core} per core, one single-threaded process per core, "
                              f"{cores} of the {host_cores} cores this box grants (cgroup CPU quota; {logical} logical CPUs visible); oracle/ipm.py with opts fast: Python driver, numpy-vectorised NLP functions, scipy "
                              f"SuperLU on an RCM-ordered KKT matrix, one factorisation per iteration), {o['n_ok']}/{o['n']} converged, "
                              f"{o['iters_mean']:.1f} iterations and {o['s_per_solve']:.2f} s per solve = {o['ms_per_iteration']:.1f} ms per "
                              f"iteration and core under this load (the reference's own logged datum: IPOPT + MUMPS 23 ms per iteration "
                              f"on a 6 408-variable robust NLP, documentation/source/getting_started.ipynb:1194-1254), parallel region "
                              f"{o['t_par']:.1f} s, {time.perf_counter() - t0:.1f} s incl. start-up and the {o['t_build']:.1f} s model build"})
    if os.environ.get("DOMPC_CPU_SAME_ALGORITHM", "1") != "0":
        t1 = time.perf_counter()
        n_he = 8 if cores <= 32 else 2
        h, err = _run_fanout("hostemu", cores, n_he, timeout=300.0)
        if h is None:
            out["same_algorithm_on_cpu"] = {"error": err}
        else:
            out["same_algorithm_on_cpu"] = {
                "value": h["value"], "unit": "MPC steps/s", "per_core": h["per_core"], "cores": cores, "ms_per_iteration": h["ms_per_iteration"],
                "sample": f"{h['n']} cold solves ({n_he} per core, {cores} single-threaded processes): the kernel text of the product compiled by "
                          f"g++ -O2 for the host (tests/_hostemu, test infrastructure), {h['n_ok']}/{h['n']} converged, {h['s_per_solve']:.2f} s per "
                          f"solve, parallel region {h['t_par']:.1f} s, {time.perf_counter() - t1:.1f} s incl. start-up",
                "note": "NOT the reference's CPU path (that is IPOPT + a general sparse LDL'); the product's structured algorithm on the host cores"}
    return out


def live_traffic(args, B: int):
    """HBM bytes of ONE launch of dompc_solve_kernel at this batch size, measured NOW: two extra passes of this script (one
    step each) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` (separate passes, MI355X_MICROARCH.md
    section HBM; counter unit KB; FETCH_SIZE counts half of the bytes of coalesced reads on gfx950 - calibrated on this
    project's 8 B/lane pattern in profiles/pmc_calibration.json: x2.0 / x1.0).  The child runs one solve launch and then one
    sweep-only launch (dompc_sweep_batch_device, same kernel symbol, mode 2): the dispatch rows are told apart by their order.
    Returns (bytes of the solve launch or None, bytes of the sweep-only launch or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, None, "rocprofv3 not found"
    tot = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="dompc_pmc_", dir="/tmp")
        cmd = [rp, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--batch", str(B), "--variant", args.variant,
               "--no-cpu-baseline", "--no-traffic", "--no-b1", "--no-variant-b", "--sweep-steps", "1", "--sweep-warmup", "0"]
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                               timeout=float(os.environ.get("DOMPC_PMC_TIMEOUT", "300")))
        except Exception as e:      # noqa: BLE001
            shutil.rmtree(d, ignore_errors=True)
            return None, None, f"{counter} pass failed: {type(e).__name__}"
        per = {}                  # dispatch id -> counter value (the child launches the solve first, then the sweep-only launch)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "dompc_solve" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    k = int(float(row.get("Dispatch_Id", 0) or 0))
                    per[k] = per.get(k, 0.0) + float(row["Counter_Value"])
        shutil.rmtree(d, ignore_errors=True)
        if not per:
            return None, None, f"{counter} pass produced no rows (rc {r.returncode}): {r.stdout[-200:]}"
        vals = [per[k] for k in sorted(per)]
        tot[counter] = (vals[0], vals[1] if len(vals) > 1 else None)
    byt = (2.0 * tot["FETCH_SIZE"][0] + tot["WRITE_SIZE"][0]) * 1024.0
    sw = None
    if tot["FETCH_SIZE"][1] is not None and tot["WRITE_SIZE"][1] is not None:
        sw = (2.0 * tot["FETCH_SIZE"][1] + tot["WRITE_SIZE"][1]) * 1024.0
    return byt, sw, (f"live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, one launch each at this batch size "
                     f"(FETCH_SIZE {tot['FETCH_SIZE'][0]:.6g} KB x2.0, WRITE_SIZE {tot['WRITE_SIZE'][0]:.6g} KB x1.0)")


def make_step_b1(mpc, ex, n_warm: int = 4) -> dict:
    """The other half of the metric: wall time of ONE MPC.make_step (B = 1, through dompc_solve: host buffers in, u0 out,
    K workgroups cooperating on the problem).  cold = first step from set_initial_guess; warm = the following steps, closed
    on the MPC's own prediction (x0 <- predicted state of scenario 0, previous solution as initial guess, optimizer.py:754-768)."""
    ps = mpc.structure
    mpc.x0 = ex.X0
    mpc.u0 = np.zeros(ps.nu)
    mpc.set_initial_guess()
    mpc.make_step(ex.X0)                         # untimed: wide-mode launch path, page-in
    mpc.x0 = ex.X0
    mpc.u0 = np.zeros(ps.nu)
    mpc._t0 = mpc._t0 * 0
    mpc.set_initial_guess()
    t0 = time.perf_counter()
    mpc.make_step(ex.X0)
    cold = (time.perf_counter() - t0) * 1e3
    ok = bool(mpc.solver_stats["success"])
    it_cold = int(mpc.solver_stats["iter_count"])
    warm, it_warm = [], []
    for _ in range(n_warm):
        i1 = ps.ix(1, 0, ps.M)
        x1 = mpc.opt_x_num.master[i1:i1 + ps.nx] * mpc._x_scaling.master
        t0 = time.perf_counter()
        mpc.make_step(x1)
        warm.append((time.perf_counter() - t0) * 1e3)
        ok = ok and bool(mpc.solver_stats["success"])
        it_warm.append(int(mpc.solver_stats["iter_count"]))
    return {"cold": cold, "warm": float(np.mean(warm)), "unit": "ms", "iters_cold": it_cold,
            "iters_warm": float(np.mean(it_warm)), "converged": ok,
            "note": "host wall time of MPC.make_step (one problem, dompc_solve, host buffers; latency-bound: device-scope "
                    "barriers between the workgroups of the problem)"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks = GPUs of this node.  Under torchrun (WORLD_SIZE in the environment) the launcher decides; "
                         "started plainly with N > 1 the script launches N ranks of itself, one per GPU")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("DOMPC_BENCH_BATCH", "16384")),
                    help="problems per GPU per step (SURVEY 8(d): B in {1, 64, 1024, 4096, 16384}; 16384 = 8 rounds over the 2048 "
                         "resident problem slots, so the uneven tail of the last round - iteration counts differ by 30 % - weighs less: "
                         "+6 % over B = 4096)")
    ap.add_argument("--variant", default="A", choices=["A", "B", "tree", "closed_loop"],
                    help="A: shipped 9x1 tree (golden-pinned); B: 3 combinations, n_robust=2; "
                         "tree: one 3^n_robust-leaf problem sharded over the ranks (strong scaling); "
                         "closed_loop: B closed loops (controller + GPU plant) resident in HBM, warm-started steps")
    ap.add_argument("--n-robust", type=int, default=5, help="--variant tree: depth of the branching part (3^n leaves)")
    ap.add_argument("--cut-level", type=int, default=0, help="--variant tree: level whose nodes are the sub-tree roots (0 = auto)")
    ap.add_argument("--shard-one-rank", action="store_true", help="--variant tree on one GPU: run the sharded code path with world = 1 (all exchanges) instead of the whole-chip mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc passes that measure roofline.traffic")
    ap.add_argument("--no-b1", action="store_true", help="skip the single-problem make_step latency")
    ap.add_argument("--no-variant-b", action="store_true", help="skip the extra `variant_b` key (SURVEY App. D's second reading of configs[3])")
    ap.add_argument("--sweep-steps", type=int, default=3, help="launches of the sweep-only kernel for roofline.sweep_only (0: skip)")
    ap.add_argument("--sweep-warmup", type=int, default=1)
    ap.add_argument("--max-soc", type=int, default=None,
                    help="measurement aid: ipopt.max_soc (default: IPOPT's 4; 0 switches the second-order correction off)")
    return ap.parse_args(argv)


def visible_devices() -> int:
    if BACKEND == "hostemu":
        return 1 << 30
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script, one process per GPU (RANK = LOCAL_RANK =
    device index, rendezvous on 127.0.0.1), the way the reference fans make_step out over processes
    (examples/tools/sampling/multiprocessing/closed_loop/mp_sampling_closed_loop_02.py:32-70).  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    n = args.gpus
    have = visible_devices()
    if have < n:
        raise SystemExit(f"bench.py --gpus {n} needs {n} devices on this node, {have} visible: one rank per GPU, nothing is "
                         f"time-shared (run it with --gpus {max(have, 1)} or under torchrun on a node that has them)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    rc, alive = 0, list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in alive:             # a rank died: the others would wait at the next barrier forever
                    q.terminate()
    return rc


class _HostTimer:
    """stand-in for a pair of HIP events on the host emulation (test plumbing)"""
    def __init__(self):
        self.t = 0.0

    def record(self, _stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, argv)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    hip = BACKEND != "hostemu"
    if hip:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the dompc IPM backend has no CPU path")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: rank {rank} wants device {local_rank}, this node has {torch.cuda.device_count()}")
        torch.cuda.set_device(local_rank)
    elif args.variant not in ("A", "B"):
        raise SystemExit("the host-emulation backend (tests) covers variants A and B only")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if hip else "gloo", rank=rank, world_size=world)

    from do_mpc_amd.examples import industrial_poly as ex
    from do_mpc_amd.solver import STATS_DTYPE
    if args.variant == "tree":
        return bench_tree(args, ex, rank, world, local_rank, dist)
    if args.variant == "closed_loop":
        return bench_closed_loop(args, ex, rank, world, local_rank, dist)
    dev = torch.device("cuda", local_rank) if hip else torch.device("cpu")
    sync = torch.cuda.synchronize if hip else (lambda: None)
    stream = torch.cuda.current_stream() if hip else None
    stream_ptr = stream.cuda_stream if hip else 0

    def build(variant, B):
        kw = {} if variant == "A" else {"n_robust": 2, "uncertainty": "paired"}
        if args.max_soc is not None:
            kw["nlpsol_opts"] = {"ipopt.max_soc": args.max_soc}
        if hip:
            return ex.build_mpc(ex.build_model(), gpu_index=local_rank, max_batch=B, **kw)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import hostemu                                       # TEST-ONLY (see BACKEND above)
        with hostemu.patched():
            return ex.build_mpc(ex.build_model(), max_batch=B, **kw)

    def resident_inputs(mpc, B):
        """synthetic inputs of this rank's shard, resident in device memory before any timed region"""
        ps = mpc.structure
        lo, hi = shard(B * world, rank, world)
        X0 = synthetic_x0_batch(B * world)[lo:hi]
        P = np.tile(mpc.opt_p_num.master, (B, 1))
        P[:, :ps.nx] = X0
        P[:, ps.p_off_p:ps.p_off_uprev] = mpc.p_fun(0.0).master
        Xi = np.zeros((B, ps.n_opt_x))
        Xi[:, :ps.off_z].reshape(B, -1, ps.nx)[:] = (X0 / mpc._x_scaling.master)[:, None, :]
        t = {"X0": torch.from_numpy(Xi).to(dev), "P": torch.from_numpy(P).to(dev),
             "lbx": torch.from_numpy(mpc._lb_opt_x.master).to(dev), "ubx": torch.from_numpy(mpc._ub_opt_x.master).to(dev),
             "lbg": torch.from_numpy(mpc._nlp_cons_lb).to(dev), "ubg": torch.from_numpy(mpc._nlp_cons_ub).to(dev),
             "X": torch.empty((B, ps.n_opt_x), dtype=torch.float64, device=dev),
             "LG": torch.empty((B, ps.n_g), dtype=torch.float64, device=dev),
             "F": torch.empty(B, dtype=torch.float64, device=dev),
             "Stats": torch.zeros(B * STATS_DTYPE.itemsize, dtype=torch.uint8, device=dev), "lo": lo, "hi": hi}
        return t

    def timed_steps(mpc, t, B, warmup, steps):
        """`warmup` untimed launches, then `steps` launches between barrier + synchronize on both sides; max over the ranks"""
        S = mpc.S

        def step():
            S.solve_batch_device(B, t["X0"].data_ptr(), t["lbx"].data_ptr(), t["ubx"].data_ptr(), t["lbg"].data_ptr(),
                                 t["ubg"].data_ptr(), t["P"].data_ptr(), t["X"].data_ptr(), 0, 0, t["LG"].data_ptr(),
                                 t["F"].data_ptr(), t["Stats"].data_ptr(), stream=stream_ptr)

        for _ in range(warmup):
            step()
        sync()
        if dist is not None:
            dist.barrier()
        mk = (lambda: torch.cuda.Event(enable_timing=True)) if hip else _HostTimer
        ev = [(mk(), mk()) for _ in range(steps)]
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record(stream)
            step()
            ev[k][1].record(stream)
        sync()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        stats = np.frombuffer(t["Stats"].cpu().numpy().tobytes(), dtype=STATS_DTYPE)
        return dt, kern_ms, stats

    B = args.batch
    mpc = build(args.variant, B)
    ps = mpc.structure
    S = mpc.S
    t = resident_inputs(mpc, B)
    dt, kern_ms, stats = timed_steps(mpc, t, B, args.warmup, args.steps)
    u0 = (t["X"][:, ps.iu(0, 0):ps.iu(0, 0) + ps.nu].cpu().numpy() * mpc._u_scaling.master)
    n_ok_all = int(stats["success"].sum())
    if dist is not None:
        tt = torch.tensor([float(n_ok_all), float(t["hi"] - t["lo"])], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        n_ok_all, n_all = int(tt[0].item()), int(tt[1].item())
    else:
        n_all = B

    if rank == 0:
        n_ok = int(stats["success"].sum())
        sweep_b = sweep_bytes_per_problem(ps)
        alg_bytes = float(stats["n_sweeps"].astype(np.float64).sum()) * sweep_b
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        # ---- the roofline north_star names literally: the Jacobian (model-evaluation) sweep ALONE.  One launch of
        # dompc_sweep_batch_device (same kernel symbol, mode 2) evaluates g, the per-edge Jacobian / Hessian blocks and their
        # condensed form at B iterates - here the B solutions of the timed batch with their multipliers; residuals out only.
        sweep_only = None
        if args.sweep_steps > 0:
            tG = torch.empty((B, ps.n_g), dtype=torch.float64, device=dev)
            mk = (lambda: torch.cuda.Event(enable_timing=True)) if hip else _HostTimer

            def sweep_launch():
                S.sweep_batch_device(B, t["X"].data_ptr(), t["LG"].data_ptr(), t["P"].data_ptr(), tG.data_ptr(), 0, stream=stream_ptr)

            for _ in range(args.sweep_warmup):
                sweep_launch()
            sync()
            evs = [(mk(), mk()) for _ in range(args.sweep_steps)]
            for a, b_ in evs:
                a.record(stream)
                sweep_launch()
                b_.record(stream)
            sync()
            sw_ms = float(np.mean([a.elapsed_time(b_) for a, b_ in evs]))
            sw_res = float(tG.abs().max().item()) if bool(np.all(mpc._nlp_cons_lb == mpc._nlp_cons_ub)) else None
            sw_ach = B * sweep_b / (sw_ms * 1e-3) / 1e9
            sweep_only = {"achieved": sw_ach, "frac": sw_ach / HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                          "kernel_ms": sw_ms, "algorithmic_bytes_per_launch": float(B * sweep_b),
                          "max_abs_residual_at_the_solutions": sw_res,      # (all rows are equalities: g(x*) = 0 up to the solver's tolerance)
                          "what": "dompc_sweep_batch_device: one model-evaluation sweep (g, per-edge Jacobian and Lagrangian-Hessian "
                                  "blocks, condensing records) of each of the B solutions, residuals copied out; dompc_solve_kernel mode 2"}
        traffic, traffic_note = None, "not measured (--no-traffic or N > 1)"
        if not args.no_traffic and world == 1 and hip:
            traffic, sw_traffic, traffic_note = live_traffic(args, B)
            if sweep_only is not None and sw_traffic:
                sweep_only["traffic"] = sw_traffic
                sweep_only["traffic_over_algorithmic"] = sw_traffic / (B * sweep_b)
        out = {
            "metric": "MPC steps/sec (make_step wall-time), industrial_poly robust multi-stage",
            "value": n_all * args.steps / dt, "unit": "MPC steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if hip else "synthetic; HOST EMULATION of the kernels (test plumbing, not a measurement)",
            "config": {"workload": f"industrial_poly robust multi-stage NMPC, variant {args.variant} "
                                   f"({'9 combos x n_robust=1' if args.variant == 'A' else '3 combos x n_robust=2'}, "
                                   f"9 scenarios, N=20, Radau deg 2)",
                       "batch_per_gpu": B, "n_opt_x": ps.n_opt_x, "n_g": ps.n_g, "edges": ps.n_edges,
                       "start": "cold (set_initial_guess semantics)", "parallelism": f"x0-batch shards x{world}",
                       "shard_of_rank0": [int(t["lo"]), int(t["hi"])], "global_batch": n_all,
                       "problem_slots": S.num_slots,
                       "code_object": {k: (os.path.basename(v) if isinstance(v, str) and v.endswith(".hsaco") else v)
                                       for k, v in getattr(S, "code_object_info", {}).items()},
                       "x0_batch": "seed 99: masses x(1 + 2 % U(-1,1)), temperatures +- 1 K U(-1,1), T_adiab recomputed "
                                   "(SURVEY 8(d) asks for 2 % relative on every state: on Kelvin temperatures that leaves the "
                                   "+-2 K reactor band and makes the robust problem infeasible, also for IPOPT)"},
            "solve": {"converged": n_ok, "of": B, "converged_all_ranks": n_ok_all, "of_all_ranks": n_all,
                      "iters_mean": float(stats["iter_count"].mean()),
                      "iters_max": int(stats["iter_count"].max()),
                      "sweeps_per_solve": float(stats["n_sweeps"].mean()), "trials_per_solve": float(stats["n_trials"].mean()),
                      "u0_first": [float(v) for v in u0[0]]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "traffic_over_algorithmic": (traffic / alg_bytes if traffic else None),
                         "kernel": "dompc_solve_kernel",
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "sweep_bytes_per_problem": sweep_b, "sweep_only": sweep_only},
        }
        if not args.no_variant_b and world == 1 and args.variant == "A":
            # SURVEY App. D's second reading of BASELINE configs[3] (3 combinations, n_robust = 2: the docs' figure), same batch
            try:
                mpc_b = build("B", B)
                tb = resident_inputs(mpc_b, B)
                dtb, kb_ms, sb = timed_steps(mpc_b, tb, B, 1, 2)
                out["variant_b"] = {"value": B * 2 / dtb, "unit": "MPC steps/s", "ms_per_step": dtb / 2 * 1e3, "steps": 2, "warmup": 1,
                                    "kernel_ms": kb_ms, "converged": int(sb["success"].sum()), "of": B,
                                    "iters_mean": float(sb["iter_count"].mean()), "edges": mpc_b.structure.n_edges, "n_g": mpc_b.structure.n_g,
                                    "workload": "industrial_poly robust multi-stage NMPC, variant B (3 combos x n_robust=2, 9 scenarios, N=20, Radau deg 2)"}
                del mpc_b, tb
            except Exception as e:      # noqa: BLE001
                out["variant_b"] = {"error": repr(e)}
        if not args.no_b1 and world == 1 and hip:
            try:
                out["make_step_ms_b1"] = make_step_b1(mpc, ex)
            except Exception as e:      # noqa: BLE001
                out["make_step_ms_b1"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def bench_closed_loop(args, ex, rank, world, local_rank, dist):
    """B closed loops per GPU: every step = one batched make_step (warm-started from the previous
    solution, like Optimizer.solve) + one batched plant step (do_mpc_amd/simulator.py), everything resident in HBM.
    The warm-up steps include the cold first solve; value = loop steps/s."""
    import torch
    from do_mpc_amd.closed_loop import BatchClosedLoop
    from do_mpc_amd.simulator import Simulator
    B = args.batch
    model = ex.build_model()
    mpc = ex.build_mpc(model, gpu_index=local_rank, max_batch=B)
    sim = Simulator(model)
    sim.set_param(t_step=float(mpc.settings.t_step), abstol=1e-10, reltol=1e-10, gpu_index=local_rank)
    pt = sim.get_p_template()
    pt["delH_R"], pt["k_0"] = 950.0, 7.0            # true plant parameters of examples/industrial_poly/template_simulator.py
    sim.set_p_fun(lambda t: pt)
    sim.setup()
    lo, hi = shard(B * world, rank, world)
    loop = BatchClosedLoop(mpc, sim, synthetic_x0_batch(B * world)[lo:hi], device=local_rank)
    iters, ok = [], True
    for _ in range(max(args.warmup, 1)):
        loop.step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = loop.step()
        iters.append(float(r["stats"]["iter_count"].mean()))
        ok = ok and bool(r["stats"]["success"].all()) and bool((r["plant_status"] == 0).all())
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", local_rank))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        ps = mpc.structure
        sweep_b = sweep_bytes_per_problem(ps)
        achieved = (float(np.mean(iters)) + 1.0) * sweep_b * B * args.steps / dt / 1e9
        print(json.dumps({
            "metric": "MPC steps/sec (make_step wall-time), industrial_poly robust multi-stage", "value": B * world * args.steps / dt,
            "unit": "MPC steps/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "industrial_poly robust multi-stage NMPC, variant A, CLOSED LOOP: warm-started make_step + "
                                   "GPU plant step per loop step, batch resident in HBM", "batch_per_gpu": B,
                       "n_opt_x": ps.n_opt_x, "n_g": ps.n_g, "edges": ps.n_edges, "start": "warm (previous solution)",
                       "parallelism": f"x0-batch shards x{world}"},
            "solve": {"converged": bool(ok), "iters_mean": float(np.mean(iters))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "dompc_solve_kernel", "kernel_ms": dt / args.steps * 1e3,
                         "sweep_bytes_per_problem": sweep_b,
                         "note": "wall time of a loop step (solver kernel + plant kernel + host bookkeeping), not a kernel-only time"},
            "cpu_baseline": None}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def bench_tree(args, ex, rank, world, local_rank, dist):
    """Strong scaling of ONE problem: the scenario tree sharded over the ranks."""
    import torch
    if dist is None and world == 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1)
    mpc = ex.build_mpc(ex.build_model(), gpu_index=local_rank, n_robust=args.n_robust, uncertainty="paired")
    ps = mpc.structure
    n_scen = ps.scenario_tree["n_scenarios"]
    # cut where there are at least 3 sub-trees per rank (balance: 27 sub-trees over 8 ranks = 4/3 per rank)
    cut = args.cut_level or next((k for k in range(1, ps.n_robust + 1) if n_scen[k] >= 3 * world), ps.n_robust)
    # one GPU: nothing to shard - the problem runs on the plain code object, its workgroups spread over the whole chip (round 5:
    # KArgs::wide_spread; --shard-one-rank runs the sharded code path with world = 1 instead, exchanges served and counted)
    sharded = world > 1 or args.shard_one_rank
    info = mpc.shard_tree(rank, world, cut_level=cut) if sharded else {"cut_level": 0, "n_cut": 0}
    t_step = []
    iters = []
    n_xchg = []
    ok = True
    for k in range(args.warmup + args.steps):
        mpc.x0 = ex.X0
        mpc.u0 = np.zeros(ps.nu)
        mpc._t0 = mpc._t0 * 0
        mpc.set_initial_guess()                     # every step is a cold start of the same problem
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        u0 = mpc.make_step(ex.X0)
        torch.cuda.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t0
        if k >= args.warmup:
            t_step.append(dt)
            iters.append(mpc.solver_stats["iter_count"])
            n_xchg.append(mpc.solver_stats.get("n_exchanges", 0))
            ok = ok and bool(mpc.solver_stats["success"])
    t = torch.tensor([sum(t_step)], dtype=torch.float64, device=torch.device("cuda", local_rank))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank == 0:
        sweep_b = sweep_bytes_per_problem(ps)
        n_sweeps = float(np.mean(iters)) + 1.0
        achieved = n_sweeps * sweep_b * args.steps / dt / 1e9
        print(json.dumps({
            "metric": "MPC steps/sec (make_step wall-time), industrial_poly robust multi-stage", "value": args.steps / dt,
            "unit": "MPC steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"industrial_poly robust multi-stage NMPC, ONE problem, {ps.S}-leaf scenario tree "
                                   f"(3 combos x n_robust={args.n_robust}, N=20, Radau deg 2)" + (", tree sharded over the ranks" if sharded else ""),
                       "n_opt_x": ps.n_opt_x, "n_g": ps.n_g, "edges": ps.n_edges, "cut_level": info["cut_level"],
                       "cut_parents": info["n_cut"], "start": "cold",
                       "parallelism": f"scenario sub-trees x{world}, RCCL all-reduce" if sharded else
                                      "one GPU, not sharded: the workgroups of the problem spread over all XCDs (whole-chip wide mode)"},
            "solve": {"converged": bool(ok), "iters_mean": float(np.mean(iters)), "u0": [float(v) for v in np.ravel(u0)],
                      "exchanges_per_solve": float(np.mean(n_xchg)), "exchanges_per_iteration": float(np.mean(n_xchg) / max(np.mean(iters), 1.0))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": achieved / (HBM_PEAK_GBS * world), "traffic": None, "kernel": "dompc_solve_kernel",
                         "kernel_ms": dt / args.steps * 1e3, "sweep_bytes_per_problem": sweep_b,
                         "note": "host wall time of make_step (includes the exchange handshakes), not a kernel-only time"},
            "cpu_baseline": None}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
