"""Throughput of the device-resident closed loop controller -> plant -> moving horizon estimator (BatchClosedLoopMHE) on the
reference's rotating-masses example: B loops advance together, three batched launches per control step.
usage: python tools/gpu_mhe_loop.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from do_mpc_amd.closed_loop import BatchClosedLoopMHE
from do_mpc_amd.examples import CASES
from do_mpc_amd.simulator import Simulator

MAX_IT = int(os.environ.get("MHE_MAX_ITER", "200"))     # (a handful of the random starts never converge without a restoration phase: bound their tail)
ex = CASES["rotating_masses"]
model = ex.build_model()
for B in [int(a) for a in sys.argv[1:]] or [64, 1024, 4096]:
    sim = Simulator(model)
    sim.set_param(t_step=0.1, abstol=1e-10, reltol=1e-10)
    pt = sim.get_p_template()
    for k in ("Theta_1", "Theta_2", "Theta_3"):
        pt[k] = 2.25e-4
    sim.set_p_fun(lambda t: pt)
    tv = sim.get_tvp_template()
    sim.set_tvp_fun(lambda t: tv)
    sim.setup()
    rng = np.random.RandomState(99)
    X0 = rng.rand(B, 8) - 0.5
    loop = BatchClosedLoopMHE(ex.build_mpc(model, max_batch=B), sim, ex.build_mhe(model, max_batch=B, nlpsol_opts={"ipopt.max_iter": MAX_IT}), X0, p_est0=1e-4)
    for _ in range(3):
        r = loop.step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 8
    ok_c = ok_e = 0
    for _ in range(n):
        r = loop.step()
        ok_c += int(r["mpc_stats"]["success"].sum()); ok_e += int(r["mhe_stats"]["success"].sum())
    dt = time.perf_counter() - t
    print(f"| rotating masses MPC (N=20) + plant + MHE (N=10, 1 parameter) | {B} | {dt / n * 1e3:.1f} ms per loop step | {B * n / dt:.0f} loop steps/s | "
          f"controller {ok_c}/{B * n} converged, {r['mpc_stats']['iter_count'].mean():.1f} iterations | estimator {ok_e}/{B * n}, {r['mhe_stats']['iter_count'].mean():.1f} (max {r['mhe_stats']['iter_count'].max()}, status codes {sorted(set(r['mhe_stats']['status'].tolist()))}) |", flush=True)
