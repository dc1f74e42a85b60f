#!/usr/bin/env python3
"""Extract numeric golden vectors from the reference's regression pickles.

Source fixtures (read-only, only present in the build container):
  /root/reference/testing/results/results_{industrial_poly,CSTR,batch_reactor,oscillatingMasses}.pkl
They are asserted at 1e-8 by the reference's own tests, e.g.
  /root/reference/testing/test_industrial_poly.py:122-139.

The pickles hold do_mpc.data.MPCData objects whose classes need CasADi to
unpickle.  CasADi is not available here, so every non-numpy class is replaced
by a stub that just records its state; all numeric payloads are plain ndarrays.

Output: tests/golden/<case>.npz  (small, committed), and tests/golden/sampling_test_fun.json - the table the reference's
sampling tool chain produces for examples/tools/sampling/regular/test_fun/sampling_test.py
(results/res_sampling_test_test_fun.pkl, compared for equality by testing/test_sampling_tools.py:55-67; plain
dicts of floats / ints / strings, written with repr-exact floats).
Run:    python tools/extract_golden.py
"""
import io
import os
import pickle
import sys

import numpy as np

REF = "/root/reference/testing/results"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

CASES = {
    "industrial_poly": "results_industrial_poly.pkl",
    "CSTR": "results_CSTR.pkl",
    "batch_reactor": "results_batch_reactor.pkl",
    "oscillating_masses": "results_oscillatingMasses.pkl",
    "rotating_masses": "results_rotatingMasses.pkl",
    "triple_tank": "results_triple_tank_ekf.pkl",
    "oscillating_masses_dae": "results_oscillatingMasses_dae.pkl",      # DAE models (`_z`): discrete, and
    "dip": "results_dip.pkl",                                           # double inverted pendulum (collocation, nl_cons, tvp)
}


class _Stub:
    def __init__(self, *a, **k):
        self._args = a
        self._kw = k

    def __setstate__(self, st):
        self._state = st

    def __reduce_ex__(self, proto):  # never re-pickled
        raise TypeError


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("numpy") or module in ("builtins", "collections", "copyreg"):
            return super().find_class(module, name)
        return type(name, (_Stub,), {"__module__": module})


def _state(obj):
    st = getattr(obj, "_state", None)
    if st is None:
        st = getattr(obj, "__dict__", {})
    return st


def main():
    os.makedirs(OUT, exist_ok=True)
    for case, fn in CASES.items():
        with open(os.path.join(REF, fn), "rb") as f:
            res = _Unpickler(io.BytesIO(f.read())).load()
        out = {}
        for who in ("mpc", "simulator", "estimator"):
            if who not in res:
                continue
            st = _state(res[who])
            for key, val in st.items():
                if isinstance(val, np.ndarray) and val.dtype != object:
                    out[f"{who}.{key}"] = val
            meta = st.get("meta_data") or st.get("_meta_data")
            if isinstance(meta, dict):
                for mk, mv in meta.items():
                    if isinstance(mv, (int, float, str, bool)):
                        out[f"{who}.meta.{mk}"] = np.array(mv)
                    elif isinstance(mv, np.ndarray) and mv.dtype != object:
                        out[f"{who}.meta.{mk}"] = mv
        path = os.path.join(OUT, case + ".npz")
        np.savez_compressed(path, **out)
        print(case, "->", path, {k: v.shape for k, v in out.items() if v.ndim > 0})
    import json
    with open(os.path.join(REF, "res_sampling_test_test_fun.pkl"), "rb") as f:
        tab = pickle.load(f)
    plain = {name: [{k: (v.item() if isinstance(v, np.generic) else v) for k, v in row.items()} for row in rows]
             for name, rows in tab.items()}
    path = os.path.join(OUT, "sampling_test_fun.json")
    with open(path, "w") as f:
        json.dump(plain, f, indent=1)
    print("sampling_test_fun ->", path, {k: len(v) for k, v in plain.items()})


if __name__ == "__main__":
    sys.exit(main())
