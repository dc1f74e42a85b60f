"""Pre-build the code objects (general + batch shape) of one example for several DOMPC_DEFS sets:
python tools/prebuild_model.py batch_reactor '{"n_horizon": 50}' "" "DOMPC_REG_GJ=0"      (DOMPC_PROFILE=1 in the environment: the profile builds)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
from do_mpc_amd import build as nb
name, kw = sys.argv[1], json.loads(sys.argv[2])
for defs in sys.argv[3:]:
    os.environ["DOMPC_DEFS"] = defs
    for _, _, header, h in g.lowered_models([(name, kw)]):
        try:
            print(repr(defs), nb.model_code_object(header, h), nb.model_code_object(header, h, batch_only=True), flush=True)
        except Exception as e:
            print(repr(defs), "BUILD FAILED", str(e)[-3000:])
