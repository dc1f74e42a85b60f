"""48 random members of the bench batch (B = 16 384): host emulation of the kernels against oracle solves - iteration counts, u0 and primal
errors.   python tools/members_hostemu.py [processes]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import multiprocessing as mp
import numpy as np

def work(i):
    import warnings; warnings.filterwarnings("ignore")
    import hostemu, bench, parity_common as pc
    from oracle import ipm
    from do_mpc_amd.examples import industrial_poly as ex
    X0 = bench.synthetic_x0_batch(16384)
    nlp = pc.oracle_nlp("industrial_poly")
    r = ipm.solve(nlp, nlp.initial_guess(X0[i]), nlp.opt_p(X0[i], np.zeros(nlp.nu)))
    with hostemu.patched():
        mpc = ex.build_mpc(ex.build_model())
    mpc.x0 = X0[i]; mpc.set_initial_guess()
    u0 = mpc.make_step(X0[i]).ravel()
    used = np.ones(mpc.structure.n_opt_x, bool); used[mpc.structure.tables["dummy_idx"]] = False
    return i, int(r["stats"]["iter_count"]), int(mpc.solver_stats["iter_count"]), float(pc.relerr(u0, nlp.u0_of(r["x"]))), float(pc.relerr(mpc.opt_x_num.master[used], r["x"][used]))

if __name__ == "__main__":
    rng = np.random.default_rng(7)
    members = sorted(int(v) for v in rng.choice(16384, 48, replace=False))
    with mp.get_context("spawn").Pool(int(sys.argv[1]) if len(sys.argv) > 1 else 8) as pool:
        for res in pool.imap_unordered(work, members):
            print(res, flush=True)
