"""Measurement aid: nothing but launches of the sweep-only kernel (dompc_sweep_batch_device, industrial_poly, B problems) -
the process tools/pmc_sweep.sh profiles: the number of edges a launch processes is known exactly (B x 180), so the counters
give instructions / busy cycles PER EDGE of the Jacobian sweep.   python tools/pmc_sweep.py [B] [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    import torch
    import bench
    from do_mpc_amd.examples import industrial_poly as ex
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda", 0)
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
    for k in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        S.sweep_batch_device(B, tX.data_ptr(), tL.data_ptr(), tP.data_ptr(), tG.data_ptr(), 0, stream=st.cuda_stream)
        b.record(st)
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    print(f"sweep-only launches B={B} edges={B * ps.n_edges} slots={S.num_slots}: " + " ".join(f"{m:.2f}" for m in ms) + " ms", flush=True)


if __name__ == "__main__":
    main()
