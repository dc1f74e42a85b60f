"""Pre-build the industrial_poly code objects of several DOMPC_DEFS sets (bench and profile variants) so that one GPU call
can A/B them:  python tools/ab_prebuild.py "DOMPC_MFMA_GJ=0" "DOMPC_EF_INLINE=1" ...   ('' = the product build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
from do_mpc_amd import build as nb
for defs in sys.argv[1:]:
    lb = "2"
    if defs.startswith("LB="):                 # "LB=3 DEF1 DEF2": resident wavefronts per SIMD the kernel is compiled for
        lb, _, defs = defs[3:].partition(" ")
    os.environ["DOMPC_LB"] = lb
    for prof in ("0", "1"):
        os.environ["DOMPC_PROFILE"] = prof
        os.environ["DOMPC_DEFS"] = defs
        for name, kw, header, h in g.lowered_models([("industrial_poly", {})]):
            try:
                print(repr(defs), "prof" + prof, nb.model_code_object(header, h), flush=True)
            except Exception as e:
                print(repr(defs), "BUILD FAILED", str(e)[-1500:])
