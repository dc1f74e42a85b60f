#!/usr/bin/env python3
"""Model hashes (= names of the gfx950 code objects) that the reference's UN-EDITED template_model.py / template_mpc.py lower
to -> tests/golden/template_hashes.json.  Needs /root/reference (build container only); the GPU box checks the in-repo cases
against this file (tests/test_gpu_parity.py), tests/test_reference_templates.py checks it against the templates here."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
from do_mpc_amd import casadi_compat, controller  # noqa: E402
from test_reference_templates import DIRS, MODEL_ARGS, MPC_ARGS, REF  # noqa: E402


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    casadi_compat.install()
    controller.HipIpmSolver = ge._NoSolver
    out = {}
    for name, d in DIRS.items():
        tm = _load(os.path.join(REF, d, "template_model.py"), f"th_{name}_tm")
        tc = _load(os.path.join(REF, d, "template_mpc.py"), f"th_{name}_tc")
        mpc = tc.template_mpc(tm.template_model(*MODEL_ARGS.get(name, ())), *MPC_ARGS.get(name, ()), silence_solver=True)
        out[name] = mpc.model_hash
        print(name, mpc.model_hash)
    # the estimator of the MHE + MPC example: template_mhe.py -> the chain problem of do_mpc_amd.estimator.MHE
    d = DIRS["rotating_masses"]
    tm = _load(os.path.join(REF, d, "template_model.py"), "th_mhe_tm")
    te = _load(os.path.join(REF, d, "template_mhe.py"), "th_mhe_te")
    mhe = te.template_mhe(tm.template_model(), silence_solver=True)
    out["rotating_masses_mhe"] = mhe._mpc.model_hash
    print("rotating_masses_mhe", mhe._mpc.model_hash)
    with open(os.path.join(ROOT, "tests", "golden", "template_hashes.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
