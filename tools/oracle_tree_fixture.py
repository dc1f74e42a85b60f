"""Oracle SOLVES of the large industrial_poly scenario trees (BASELINE configs[4]: 3^5 = 243 leaves, 218 700 variables; and the 81-leaf
tree in between) -> tests/golden/oracle_tree{81,243}.npz.  The oracle (oracle/ipm.py: IPOPT's algorithm on scipy's sparse LU) needs
minutes for these, far too long for the test suite, so its cold solve from the example's x0 is stored once: iteration and
regularisation counts, objective, u0, the complete primal solution and the multipliers of g as float64 (a few MB, compressed).
    python tools/oracle_tree_fixture.py 4 5
tests/parity_common.py: check_big_tree_against_stored_oracle_solve compares the GPU solve with it; a `slow` CPU test re-runs the oracle
and checks that the stored file is what it produces."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity_common as pc
from oracle import ipm
from do_mpc_amd.examples import industrial_poly as ex


def solve(n_robust):
    nlp = pc.oracle_nlp("industrial_poly", n_robust=n_robust, p_values=pc.PAIRED_P)
    t0 = time.time()
    r = ipm.solve(nlp, nlp.initial_guess(ex.X0), nlp.opt_p(ex.X0, np.zeros(nlp.nu)), opts={"fast": True})
    return nlp, r, time.time() - t0


if __name__ == "__main__":
    for a in sys.argv[1:]:
        n_robust = int(a)
        nlp, r, dt = solve(n_robust)
        st = r["stats"]
        leaves = 3 ** n_robust
        out = os.path.join(ROOT, "tests", "golden", "oracle_tree%d.npz" % leaves)
        np.savez_compressed(out, x=r["x"], lam_g=r["lam_g"], u0=nlp.u0_of(r["x"]), x0=ex.X0, iter_count=st["iter_count"], n_reg=st["n_reg"],
                            success=st["success"], f=r["f"], n_opt_x=nlp.n_opt_x, n_g=nlp.n_g)
        print("n_robust %d: %d leaves, n_opt_x %d, n_g %d: %d iterations, n_reg %d, success %s, %.1f s -> %s (%.1f MB)"
              % (n_robust, leaves, nlp.n_opt_x, nlp.n_g, st["iter_count"], st["n_reg"], st["success"], dt, out, os.path.getsize(out) / 1e6), flush=True)
