"""industrial_poly with 80 collocation unknowns per interval on the GPU: status, iterations, comparison with the oracle (debug aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import parity_common as pc
from do_mpc_amd.examples import industrial_poly as ex
kw = dict(pc.BIG_INTERVAL)
for wide in (None, "1", "2", "4"):
    if wide is None:
        os.environ.pop("DOMPC_WIDE", None)
    else:
        os.environ["DOMPC_WIDE"] = wide
    mpc = ex.build_mpc(ex.build_model(), **kw)
    x0 = pc.golden("industrial_poly")["mpc._x"][0]
    mpc.x0 = x0; mpc.set_initial_guess()
    u0 = mpc.make_step(x0)
    st = mpc.solver_stats
    print("DOMPC_WIDE", wide, {k: st[k] for k in ("success", "return_status", "iter_count", "n_reg") if k in st}, u0.ravel(), flush=True)
    tr = mpc.S.trace(12)
    print(np.array2string(tr[:12, :6], precision=4, max_line_width=200), flush=True)
