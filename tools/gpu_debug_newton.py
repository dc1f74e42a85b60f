#!/usr/bin/env python3
"""Debug helper: compare the device Newton step pieces with the oracle, by variable class."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_common as pc
from do_mpc_amd.examples import CASES
from oracle import ipm

name = sys.argv[1] if len(sys.argv) > 1 else "industrial_poly"
ex = CASES[name]
mpc = ex.build_mpc(ex.build_model())
nlp = pc.oracle_nlp(name)
p = nlp.opt_p(ex.X0, np.zeros(nlp.nu))
r = ipm.solve(nlp, nlp.initial_guess(ex.X0), p, opts=dict(max_iter=6))
x, lam, mu = r["x"], r["lam_g"] * r["stats"]["obj_scaling"], r["stats"]["mu"]
lb, ub = nlp.lbx.copy(), nlp.ubx.copy()
hl, hu = np.isfinite(lb), np.isfinite(ub)
lb[hl] -= 1e-8 * np.maximum(1, np.abs(lb[hl])); ub[hu] += 1e-8 * np.maximum(1, np.abs(ub[hu]))
dl, du = np.where(hl, x - lb, 1.0), np.where(hu, ub - x, 1.0)
zl, zu = np.where(hl, mu / dl, 0.0), np.where(hu, mu / du, 0.0)
for rep in range(2):
    dx, dlam, rd, c = mpc.S.debug_newton_step(x, lam, zl, zu, lb, ub, nlp.lbg, nlp.ubg, p, mu, 0.0)
    A = nlp.jac(x, p); gf = nlp.grad(x, p)
    rd_o = gf + A.T @ lam - zl + zu
    ps = mpc.structure
    err = np.abs(rd - rd_o)
    print("rep", rep, "max rd err", err.max(), "at", err.argmax(), "off_u", ps.off_u, "n", ps.n_opt_x)
    bad = np.where(err > 1e-8 * max(1, np.abs(rd_o).max()))[0]
    print(" n bad", len(bad), bad[:20])
    if len(bad):
        M1 = ps.M + 1
        for g in bad[:10]:
            if g < ps.off_u:
                blk = g // ps.nx; k = blk // (ps.S * M1); s = (blk // M1) % ps.S; slot = blk % M1
                print("   x idx", g, "k", k, "s", s, "slot", slot, "state", g % ps.nx, rd[g], rd_o[g])
            else:
                print("   u idx", g, rd[g], rd_o[g])
    print(" c err", np.abs(c - (nlp.g(x, p) - nlp.lbg)).max(), "dx nan", np.isnan(dx).sum())
