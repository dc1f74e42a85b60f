"""Random starts (x0 of the example x (1 +- 3 %), ten per case) of CSTR (9-scenario tree), batch_reactor and oscillating masses: host emulation of the
kernels against oracle solves - iteration counts, u0 and primal errors.   python tools/random_starts_hostemu.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import multiprocessing as mp
import numpy as np

def work(args):
    import warnings; warnings.filterwarnings("ignore")
    name, seed = args
    import hostemu, parity_common as pc
    from oracle import ipm
    from do_mpc_amd.examples import CASES
    ex = CASES[name]
    rng = np.random.default_rng(seed)
    x0 = ex.X0 * (1.0 + 0.03 * rng.uniform(-1, 1, ex.X0.size))
    nlp = pc.oracle_nlp(name)
    r = ipm.solve(nlp, nlp.initial_guess(x0), nlp.opt_p(x0, np.zeros(nlp.nu)))
    with hostemu.patched():
        mpc = ex.build_mpc(ex.build_model())
    mpc.x0 = x0; mpc.set_initial_guess()
    u0 = mpc.make_step(x0).ravel()
    used = np.ones(mpc.structure.n_opt_x, bool); used[mpc.structure.tables["dummy_idx"]] = False
    return name, seed, bool(r["stats"]["success"]), int(r["stats"]["iter_count"]), bool(mpc.solver_stats["success"]), int(mpc.solver_stats["iter_count"]), float(pc.relerr(u0, nlp.u0_of(r["x"]))), float(pc.relerr(mpc.opt_x_num.master[used], r["x"][used]))

if __name__ == "__main__":
    jobs = [(n, s) for n in ("CSTR", "batch_reactor", "oscillating_masses") for s in range(10)]
    with mp.get_context("spawn").Pool(8) as pool:
        for res in pool.imap_unordered(work, jobs):
            print(res, flush=True)
