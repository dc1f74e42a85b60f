"""Where the host wall time of MPC.make_step goes (one problem): Python around the solver call, the C ABI call, the kernel.
python tools/gpu_make_step_overhead.py [case] [n]"""
import os, sys, time, cProfile, pstats, io
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from do_mpc_amd.examples import CASES
name = sys.argv[1] if len(sys.argv) > 1 else "industrial_poly"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ex = CASES[name]
mpc = ex.build_mpc(ex.build_model())
mpc.x0 = ex.X0; mpc.set_initial_guess()
S = mpc.S
t_call = []
orig = S.__class__.__call__
def timed(self, *a, **k):
    t0 = time.perf_counter(); r = orig(self, *a, **k); t_call.append(time.perf_counter() - t0); return r
S.__class__.__call__ = timed
x = ex.X0.copy()
mpc.make_step(x)
t_tot = []
pr = cProfile.Profile()
for k in range(n):
    mpc.set_initial_guess()            # cold every time
    t0 = time.perf_counter()
    if k >= n // 2: pr.enable()
    mpc.make_step(x)
    if k >= n // 2: pr.disable()
    t_tot.append(time.perf_counter() - t0)
t_tot, t_c = np.array(t_tot) * 1e3, np.array(t_call[1:]) * 1e3
print("make_step %.2f ms (min %.2f)   solver call %.2f ms (min %.2f)   python around it %.2f ms   iterations %d" % (
    np.median(t_tot), t_tot.min(), np.median(t_c), t_c.min(), np.median(t_tot - t_c), mpc.solver_stats["iter_count"]))
st = mpc.S.stats()
print({k: v for k, v in st.items() if k.startswith("t_")})
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
