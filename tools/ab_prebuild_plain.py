"""Pre-build the industrial_poly code objects (general + batch shape, no phase clocks) of several DOMPC_DEFS sets:
python tools/ab_prebuild_plain.py "" "DOMPC_REPEAT_PHASE=1" ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
from do_mpc_amd import build as nb
for defs in sys.argv[1:]:
    os.environ["DOMPC_DEFS"] = defs
    for name, kw, header, h in g.lowered_models([("industrial_poly", {})]):
        try:
            print(repr(defs), nb.model_code_object(header, h), nb.model_code_object(header, h, batch_only=True), flush=True)
        except Exception as e:
            print(repr(defs), "BUILD FAILED", str(e)[-1500:])
