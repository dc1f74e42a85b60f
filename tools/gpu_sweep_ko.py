"""Measurement aid: what does each piece of the edge sweep cost in THROUGHPUT (not in one wavefront's own clock)?
Code objects built with -DDOMPC_KO=<mask> leave pieces of the sweep out (wrong results!); the sweep-only launch
(dompc_sweep_batch_device, B iterates) is timed for each, at the full and at half the resident problem slots.
   python tools/gpu_sweep_ko.py [B]        (build the variants first: python tools/gpu_sweep_ko.py --build)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

VARIANTS = [("full", ""), ("no factorisation", "DOMPC_KO=1"), ("no condensing", "DOMPC_KO=2"), ("no record stores", "DOMPC_KO=4"),
            ("no model evaluation", "DOMPC_KO=8"), ("no per-variable loads", "DOMPC_KO=16"), ("no MO staging", "DOMPC_KO=48"),
            ("no factor+condense", "DOMPC_KO=3"), ("loads+stores only", "DOMPC_KO=11"), ("nothing but the loop", "DOMPC_KO=63")]


if os.environ.get("DOMPC_KO_VARIANTS"):      # "label:defs;label:defs" - other build switches timed the same way
    VARIANTS = [tuple(v.split(":", 1)) for v in os.environ["DOMPC_KO_VARIANTS"].split(";")]


def build_all():
    import __graft_entry__ as g
    from do_mpc_amd import build as nb
    from concurrent.futures import ThreadPoolExecutor
    (name, kw, header, h), = g.lowered_models([("industrial_poly", {})])

    def one(defs):
        env_defs = defs
        os.environ["DOMPC_DEFS"] = env_defs      # (threads share the environment: built one after the other below)
        return nb.model_code_object(header, h)
    for label, defs in VARIANTS:
        if defs:
            print(label, one(defs), flush=True)
    os.environ.pop("DOMPC_DEFS", None)


def main():
    import torch
    import bench
    from do_mpc_amd.examples import industrial_poly as ex
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    dev = torch.device("cuda", 0)
    rows = []
    for slots in (0, 1024):
        for label, defs in VARIANTS:
            if defs:
                os.environ["DOMPC_DEFS"] = defs
            else:
                os.environ.pop("DOMPC_DEFS", None)
            if slots:
                os.environ["DOMPC_SLOTS"] = str(slots)
            else:
                os.environ.pop("DOMPC_SLOTS", None)
            mpc = ex.build_mpc(ex.build_model(), max_batch=B)
            ps, S = mpc.structure, mpc.S
            X0 = bench.synthetic_x0_batch(B)
            P = np.tile(mpc.opt_p_num.master, (B, 1))
            P[:, :ps.nx] = X0
            P[:, ps.p_off_p:ps.p_off_uprev] = mpc.p_fun(0.0).master
            Xi = np.zeros((B, ps.n_opt_x))
            Xi[:, :ps.off_z].reshape(B, -1, ps.nx)[:] = (X0 / mpc._x_scaling.master)[:, None, :]
            tX, tP = torch.from_numpy(Xi).to(dev), torch.from_numpy(P).to(dev)
            tL = torch.zeros((B, ps.n_g), dtype=torch.float64, device=dev)
            tG = torch.empty((B, ps.n_g), dtype=torch.float64, device=dev)
            st = torch.cuda.current_stream()
            ms = []
            for k in range(4):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(st)
                S.sweep_batch_device(B, tX.data_ptr(), tL.data_ptr(), tP.data_ptr(), tG.data_ptr(), 0, stream=st.cuda_stream)
                b.record(st)
                torch.cuda.synchronize()
                ms.append(a.elapsed_time(b))
            rows.append((S.num_slots, label, min(ms[1:])))
            print(f"slots {S.num_slots:5d}  {label:24s} {min(ms[1:]):8.2f} ms", flush=True)
            del mpc, S
    os.environ.pop("DOMPC_DEFS", None)
    os.environ.pop("DOMPC_SLOTS", None)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build_all()
    else:
        main()
