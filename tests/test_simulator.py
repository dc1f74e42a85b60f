"""Batched plant integrator (do_mpc_amd/simulator.py, csrc/dompc_plant.hip) on the host emulation
of its kernel: against scipy's Radau at 1e-11 on the same right-hand side (tests/plant.py), batch semantics, limits, and
the reference's closed-loop tests with this integrator as the plant."""
import hostemu_build
import numpy as np
import pytest

import hostemu
import simulator_common as sc
from do_mpc_amd import Model
from do_mpc_amd.examples import CASES
from do_mpc_amd.simulator import Simulator
from test_closed_loop import CL_RTOL, run_closed_loop


@pytest.mark.parametrize("name", ["oscillating_masses", "batch_reactor", "CSTR", "industrial_poly"])
def test_make_step_matches_scipy_radau(name):
    sc.check_against_scipy(name, hostemu=True)


@pytest.mark.parametrize("name", ["oscillating_masses", "CSTR", "industrial_poly"])
def test_batch_semantics(name):
    sc.check_batch(name, hostemu=True)


def test_step_limit_is_reported_not_spun_on():
    sim = sc.make_simulator("industrial_poly", hostemu=True, max_steps=5)
    r = sim.make_step_batch(CASES["industrial_poly"].X0[None, :], np.array(sc.U_TEST["industrial_poly"]))
    assert r["status"][0] == 1 and r["n_steps"][0] == 5
    sim.x0 = CASES["industrial_poly"].X0
    with pytest.raises(RuntimeError, match="did not reach t_step"):
        sim.make_step(np.array(sc.U_TEST["industrial_poly"]).reshape(-1, 1))


def test_measurement_function_and_noise_inputs():
    m = Model("continuous")
    x = m.set_variable("_x", "x", (2, 1))
    u = m.set_variable("_u", "u")
    m.set_rhs("x", -x * np.array([[1.0], [2.0]]) + u, process_noise=True)
    m.set_meas("y0", x[0] + 2.0 * x[1], meas_noise=True)
    m.setup()
    sim = Simulator(m)
    sim.set_param(t_step=0.1)
    hdr = sim._lower()
    from do_mpc_amd import build
    sim.setup(_lib_path=hostemu_build.plant_hostemu_library(hdr, hdr.rsplit('PLANT_MODEL_HASH "', 1)[1].split('"')[0], sc.OUT), _code_object="")
    sim.x0 = np.array([1.0, 1.0])
    y = sim.make_step(np.array([[0.5]]), v0=np.array([[0.01]]), w0=np.array([[0.2], [0.0]])).ravel()
    # x' = -a x + u + w  ->  x(t) = c + (x0 - c) exp(-a t),  c = (u + w) / a
    x0n = 0.7 + 0.3 * np.exp(-0.1)
    x1n = 0.25 + 0.75 * np.exp(-0.2)
    assert np.allclose(sim.x0.master, [x0n, x1n], rtol=1e-11, atol=0)
    assert abs(y[0] - (x0n + 2.0 * x1n + 0.01)) < 1e-11


def test_algebraic_states_are_solved_inside_the_right_hand_side():
    """x' = -x + z, 0 = z - 2 x  ->  x' = x: the algebraic state is eliminated by Newton's method in every evaluation"""
    m = Model("continuous")
    x = m.set_variable("_x", "x")
    z = m.set_variable("_z", "z")
    m.set_rhs("x", -x + z)
    m.set_alg("a", z - 2.0 * x)
    m.setup()
    sim = Simulator(m)
    sim.set_param(t_step=0.1)
    hdr = sim._lower()
    from do_mpc_amd import build
    sim.setup(_lib_path=hostemu_build.plant_hostemu_library(hdr, hdr.rsplit('PLANT_MODEL_HASH "', 1)[1].split('"')[0], sc.OUT), _code_object="")
    sim.x0 = np.array([1.5])
    sim.make_step(np.zeros((0, 1)))
    assert abs(sim.x0.master[0] - 1.5 * np.exp(0.1)) < 1e-11


@pytest.mark.parametrize("name", ["oscillating_masses_dae", "dip"])
def test_dae_plants_against_scipy(name):
    """the two DAE examples of the reference (discrete with algebraic states; double inverted pendulum: accelerations as
    algebraic states of the Euler-Lagrange equations) against tests/plant.py (scipy Radau, z by Newton with a finite-difference Jacobian)"""
    sc.check_against_scipy(name, hostemu=True)


@pytest.mark.parametrize("name", ["oscillating_masses", "batch_reactor", "CSTR", "industrial_poly", "oscillating_masses_dae", "dip"])
def test_closed_loop_with_this_plant_reproduces_the_reference_trajectory(name):
    """(DAE examples: states of the double-inverted-pendulum loop 2e-8 from the IDAS run of the reference over three steps)"""
    from test_closed_loop import CL_STEPS

    def make_mpc(n):
        ex = CASES[n]
        with hostemu.patched():
            return ex.build_mpc(ex.build_model())
    wu, wx = run_closed_loop(make_mpc, name, steps=CL_STEPS.get(name, 5), make_plant=sc.closed_loop_plant(hostemu=True))
    assert wu < CL_RTOL and wx < CL_RTOL


@pytest.mark.parametrize("name", ["batch_reactor", "CSTR", "industrial_poly", "dip"])
def test_implicit_method_matches_scipy_radau(name):
    sc.check_implicit_against_scipy(name, hostemu=True)


def test_stiff_plant_switches_to_the_implicit_method():
    sc.check_stiff_plant(hostemu=True)


def test_newton_start_of_algebraic_states_does_not_depend_on_call_history():
    sc.check_z_start_does_not_depend_on_call_history(hostemu=True)
