"""Fixture for the Newton-direction test on the LAST barrier level (tests/parity_common.check_newton_step_at_late_iterate): the oracle's
iterate 54 of member 2 048 of the bench batch (industrial_poly, x0 of bench.synthetic_x0_batch) - barrier parameter 2.5e-9, Sigma of the
active bounds up to 2e11: the point where the multiplier steps  d nu = P dx + p  of the structured solve left 1.9e-7 in the x rows of the
linear system.  Writes tests/golden/oracle_late_iterate_ip2048.npz (one oracle solve, about a minute).
   python tools/late_iterate_fixture.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import bench
import parity_common as pc
from oracle import ipm

MEMBER, ITER = 2048, 54
x0 = bench.synthetic_x0_batch(4096)[MEMBER]
nlp = pc.oracle_nlp("industrial_poly")
p = nlp.opt_p(x0, np.zeros(nlp.nu))
tr = []
r = ipm.solve(nlp, nlp.initial_guess(x0), p, trace=tr)
t = tr[ITER]
assert abs(t["sf"] - 1.0) < 1e-12 and t["mu"] < 1e-8
out = os.path.join(R, "tests", "golden", "oracle_late_iterate_ip2048.npz")
np.savez_compressed(out, x=t["x"], y=t["y"], zl=t["zl"], zu=t["zu"], mu=t["mu"], delta=max(1e-20, t["delta_w_last"] / 3.0), x0=x0,
                    member=MEMBER, iteration=ITER, iter_count=int(r["stats"]["iter_count"]))
print(out, "mu %.3e" % t["mu"], "iterations of the solve", r["stats"]["iter_count"])
