"""The two slow solves the round-3 review attributed to the missing restoration phase, on the host emulation of the kernels (iteration
trace of the device driver): the straggler of the cold estimator batch (problem 3398 of tools/gpu_config_table.py's B = 4096 batch) and
the full-horizon kite example (N = 80) - each WITHOUT and WITH IPOPT's watchdog procedure (ipopt.watchdog_shortened_iter_trigger = 0 / 10).
Prints the line-search statistics that decide whether IPOPT's restoration phase would have been entered (alpha below alpha_min = failed
line search: it was not - the steps were accepted, at 2^-10 of their length) and an excerpt of the traces.
Usage: python tools/hostemu_crawl_traces.py > profiles/r04_crawl_traces.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hostemu
from do_mpc_amd.examples import CASES

np.set_printoptions(linewidth=200, precision=3)
KEYS = ("success", "status", "iter_count", "n_reg", "n_ls_fail", "n_trials", "n_soc", "n_watchdog", "n_sweeps", "inf_pr", "inf_du", "obj")


def show(name, st, tr):
    print(f"== {name}")
    print("  ", {k: (float(st[k]) if ("inf" in k or k == "obj") else int(st[k])) for k in KEYS})
    print(f"   trial points per iteration {st['n_trials'] / max(1, st['iter_count']):.1f}, line searches that ran out of step sizes: {int(st['n_ls_fail'])}")
    n = int(st["iter_count"])
    print("   columns: it, mu, E0, inf_pr, inf_du, alpha (negative: not accepted), delta_w, objective")
    for row in list(tr[:26]) + list(tr[26:n:max(1, n // 30)]) + list(tr[max(26, n - 6):n]):
        print("  ", row)


# 1. the estimator straggler: window 3 of the reference's stored run, perturbed measurements (seed 5, sample 3398 of 4096), cold start
ex = CASES["rotating_masses"]
g = np.load(os.path.join(ROOT, "tests", "golden", "rotating_masses.npz"))
OP = g["estimator.opt_p_num"]
with hostemu.patched():
    mhe = ex.build_mhe(ex.build_model(), max_batch=1)
    mhe0 = ex.build_mhe(ex.build_model(), max_batch=1, nlpsol_opts={"ipopt.watchdog_shortened_iter_trigger": 0})
rng = np.random.default_rng(5)
for B in (1, 64, 1024, 4096):
    idx = 1 + np.arange(B) % 4
    P_ref = OP[idx].copy()
    P_ref[:, mhe._po_y:] += 1e-3 * rng.standard_normal((B, P_ref.shape[1] - mhe._po_y))
b = 3398
init0 = np.zeros(mhe.n_opt_x); init0[mhe._o_p:] = 1e-4
for label, e_ in (("without the watchdog", mhe0), ("with the watchdog (default)", mhe)):
    mpc = e_._mpc
    r = e_.S.solve_batch(e_._to_chain(init0[None, :]), mpc._lb_opt_x.master, mpc._ub_opt_x.master, mpc._nlp_cons_lb, mpc._nlp_cons_ub,
                         e_._p_to_chain(P_ref[b:b + 1]))
    show("estimator batch (rotating masses, N = 10), cold, sample 3398, " + label, r["stats"][0], e_.S.trace(3000))

# 2. kite, full horizon
ex = CASES["kite"]
for label, trig in (("without the watchdog", 0), ("with the watchdog (default)", 10)):
    with hostemu.patched():
        mpc = ex.build_mpc(ex.build_model(), n_horizon=80, nlpsol_opts={"ipopt.watchdog_shortened_iter_trigger": trig})
    mpc.x0 = ex.X0; mpc.set_initial_guess(); mpc.make_step(ex.X0)
    st = {k: mpc.solver_stats[k] for k in KEYS if k in mpc.solver_stats}
    st["status"] = 0 if st["success"] else 2
    show("kite N = 80, cold, " + label, st, mpc.S.trace(3100))
