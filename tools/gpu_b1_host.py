"""Where the host wall time of ONE make_step goes (industrial_poly, B = 1): Python surface vs the C ABI call (`t_wall_total` = the
whole dompc_solve incl. staging copies) vs the kernel alone (HIP events around a device-resident call).
Usage: python tools/gpu_b1_host.py [case]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from do_mpc_amd.examples import CASES

name = sys.argv[1] if len(sys.argv) > 1 else "industrial_poly"
ex = CASES[name]
mpc = ex.build_mpc(ex.build_model())
x0 = ex.X0


def cold():
    mpc.x0 = x0
    mpc.set_initial_guess()
    t = time.perf_counter()
    mpc.make_step(x0)
    return (time.perf_counter() - t) * 1e3, mpc.solver_stats["t_wall_total"] * 1e3, mpc.solver_stats["iter_count"]


rows = [cold() for _ in range(8)][2:]
print(f"{name} cold make_step: wall {np.median([r[0] for r in rows]):.2f} ms, inside dompc_solve {np.median([r[1] for r in rows]):.2f} ms, iterations {rows[-1][2]}")
warm = []
for _ in range(8):
    t = time.perf_counter()
    mpc.make_step(x0)
    warm.append(((time.perf_counter() - t) * 1e3, mpc.solver_stats["t_wall_total"] * 1e3, mpc.solver_stats["iter_count"]))
print(f"{name} warm make_step: wall {np.median([r[0] for r in warm]):.2f} ms, inside dompc_solve {np.median([r[1] for r in warm]):.2f} ms, iterations {warm[-1][2]}")
# the kernel alone: device-resident call, HIP events
ps, S = mpc.structure, mpc.S
dev = torch.device("cuda", 0)
t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)      # noqa: E731
mpc.x0 = x0
mpc.set_initial_guess()
P = mpc.opt_p_num.master.copy()
P[:ps.nx] = x0
X = t_(mpc.opt_x_num.master[None, :]); Pt = t_(P[None, :])
lbx, ubx, lbg, ubg = t_(mpc._lb_opt_x.master), t_(mpc._ub_opt_x.master), t_(mpc._nlp_cons_lb), t_(mpc._nlp_cons_ub)
out = torch.empty((1, ps.n_opt_x), dtype=torch.float64, device=dev)
f = torch.empty(1, dtype=torch.float64, device=dev)
from do_mpc_amd.solver import STATS_DTYPE
st = torch.zeros(STATS_DTYPE.itemsize, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream()
ms = []
for _ in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    S.solve_batch_device(1, X.data_ptr(), lbx.data_ptr(), ubx.data_ptr(), lbg.data_ptr(), ubg.data_ptr(), Pt.data_ptr(), out.data_ptr(), 0, 0, 0,
                         f.data_ptr(), st.data_ptr(), stream=stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
it = np.frombuffer(st.cpu().numpy().tobytes(), dtype=STATS_DTYPE)["iter_count"][0]
print(f"{name} cold, device-resident call (HIP events: 2 memsets + kernel): {np.median(ms[2:]):.2f} ms, iterations {it}")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    mpc.make_step(x0)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
