"""The reference's own closed-loop tests (testing/test_*.py: 5 steps of mpc.make_step -> simulator.make_step,
results compared with testing/results/results_*.pkl) repeated on this backend: the MPC runs the product kernels
(host emulation here, the HIP path in tests/test_gpu_parity.py), the plant is tests/plant.py (scipy Radau in
place of CVODES).  Stated tolerance: inputs and states relative 2e-6 of max(1,|value|) - the goldens are IPOPT
iterates at mu = 9.09e-10 (see parity_common.py) and the trajectories accumulate the differences over 5 steps."""
import numpy as np
import pytest

import hostemu
import parity_common as pc
import plant
from do_mpc_amd.examples import CASES

CL_RTOL = 2e-6


def run_closed_loop(make_mpc, name, steps=5, make_plant=None):
    """make_plant(name, model, t_step) -> callable (x, u) -> x_next; default: tests/plant.py (scipy Radau)"""
    ex = CASES[name]
    mpc = make_mpc(name)
    g = pc.golden(name)
    U, Xs = g["mpc._u"], g["mpc._x"]
    p = plant.p_vector(mpc.model, plant.PLANT_P[name])
    x = np.array(Xs[0], float)
    mpc.x0 = x
    mpc.set_initial_guess()
    t_step = float(mpc.settings.t_step)
    step = make_plant(name, mpc.model, t_step) if make_plant else (lambda x_, u_: plant.plant_step(mpc.model, x_, u_, p, t_step))
    worst_u = worst_x = 0.0
    for k in range(steps):
        assert pc.relerr(x, Xs[k]) < CL_RTOL, (name, k, x, Xs[k])
        u0 = mpc.make_step(x).ravel()
        assert mpc.solver_stats["success"], (name, k, mpc.solver_stats)
        worst_u = max(worst_u, pc.relerr(u0, U[k]))
        assert worst_u < CL_RTOL, (name, k, u0, U[k])
        x = step(x, u0)
        if k + 1 < len(Xs):
            worst_x = max(worst_x, pc.relerr(x, Xs[k + 1]))
    return worst_u, worst_x


CL_STEPS = {"dip": 3}          # (the swing-up solves take 100+ iterations each on the host emulation)


@pytest.mark.parametrize("name", ["oscillating_masses", "batch_reactor", "CSTR", "industrial_poly", "oscillating_masses_dae", "dip"])
def test_closed_loop_reproduces_the_reference_trajectory(name):
    def make_mpc(n):
        ex = CASES[n]
        with hostemu.patched():
            return ex.build_mpc(ex.build_model())
    wu, wx = run_closed_loop(make_mpc, name, steps=CL_STEPS.get(name, 5))
    assert wu < CL_RTOL and wx < CL_RTOL
