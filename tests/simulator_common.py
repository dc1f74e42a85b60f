"""Shared helpers of the plant-integrator tests (host emulation on CPU, HIP path under -m gpu)."""
import hostemu_build
import os

import numpy as np

import plant
from do_mpc_amd import build
from do_mpc_amd.examples import CASES
from do_mpc_amd.simulator import Simulator

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_hostemu")
T_STEP = {"CSTR": 0.005, "batch_reactor": 1.0, "industrial_poly": 50.0 / 3600.0, "oscillating_masses": 0.5,
          "oscillating_masses_dae": 0.5, "dip": 0.04}
U_TEST = {"CSTR": [20.0, -3000.0], "batch_reactor": [0.05], "industrial_poly": [20000.0, 350.0, 350.0], "oscillating_masses": [0.3],
          "oscillating_masses_dae": [0.3], "dip": [1.5]}


def make_simulator(name, hostemu=True, model=None, **params):
    """the reference's template_simulator.py of the example: t_step, abstol = reltol = 1e-10, true plant parameters"""
    ex = CASES[name]
    m = model or ex.build_model()
    sim = Simulator(m)
    sim.set_param(integration_tool="cvodes", abstol=1e-10, reltol=1e-10, t_step=T_STEP[name])
    sim.set_param(**params)
    if m.n_p:
        pt = sim.get_p_template()
        for k, v in plant.PLANT_P[name].items():
            pt[k] = v
        sim.set_p_fun(lambda t: pt)
    if m.n_tvp:
        tv = sim.get_tvp_template()
        sim.set_tvp_fun(lambda t: tv)
    if hostemu:
        hdr = sim._lower()
        h = hdr.rsplit('PLANT_MODEL_HASH "', 1)[1].split('"')[0]
        sim.setup(_lib_path=hostemu_build.plant_hostemu_library(hdr, h, OUT), _code_object="")
    else:
        sim.setup()
    return sim


def check_against_scipy(name, hostemu):
    ex = CASES[name]
    sim = make_simulator(name, hostemu)
    m = sim.model
    p = plant.p_vector(m, plant.PLANT_P[name])
    sim.x0 = ex.X0
    x = ex.X0.copy()
    for k in range(3):                                     # three consecutive intervals through make_step
        u = np.array(U_TEST[name]) * (1.0 + 0.1 * k)
        y = sim.make_step(u.reshape(-1, 1)).ravel()
        x = plant.plant_step(m, x, u, p, T_STEP[name])
        assert np.max(np.abs(y - x) / np.maximum(1.0, np.abs(x))) < 1e-9, (name, k, y, x)
        assert np.allclose(sim.x0.master, y, rtol=0, atol=0)
    assert abs(float(sim.t0[0]) - 3 * T_STEP[name]) < 1e-12


def check_batch(name, hostemu, B=9):
    ex = CASES[name]
    sim = make_simulator(name, hostemu)
    m = sim.model
    rng = np.random.default_rng(5)
    X = ex.X0[None, :] * (1.0 + 0.01 * rng.uniform(-1, 1, size=(B, m.n_x)))
    U = np.array(U_TEST[name])[None, :] * (1.0 + 0.2 * rng.uniform(-1, 1, size=(B, m.n_u)))
    r = sim.make_step_batch(X, U)                          # per-sample inputs, shared parameters
    assert r["x"].shape == (B, m.n_x) and (r["status"] == 0).all() and (r["n_steps"] >= 1).all()
    p = plant.p_vector(m, plant.PLANT_P[name])
    for b in (0, B - 1):
        ref = plant.plant_step(m, X[b], U[b], p, T_STEP[name])
        assert np.max(np.abs(r["x"][b] - ref) / np.maximum(1.0, np.abs(ref))) < 1e-9
    r1 = sim.make_step_batch(X, U[0])                      # one input row shared by the batch
    one = sim.make_step_batch(X[3:4], U[0])
    assert np.array_equal(r1["x"][3], one["x"][0])         # a sample does not depend on its neighbours in the batch
    assert np.array_equal(r["y"], r["x"])                  # state feedback: y = x
    if m.n_p:                                              # per-sample parameters
        P = np.tile(p, (B, 1)) * (1.0 + 0.05 * rng.uniform(-1, 1, size=(B, m.n_p)))
        rp = sim.make_step_batch(X, U, P=P)
        ref = plant.plant_step(m, X[2], U[2], P[2], T_STEP[name])
        assert np.max(np.abs(rp["x"][2] - ref) / np.maximum(1.0, np.abs(ref))) < 1e-9


def closed_loop_plant(hostemu):
    def make_plant(name, model, t_step):
        sim = make_simulator(name, hostemu, model=model)

        def step(x, u):
            sim.x0 = x
            return sim.make_step(np.asarray(u, float).reshape(-1, 1)).ravel()
        return step
    return make_plant


def check_implicit_against_scipy(name, hostemu, tol=2e-9):
    """the implicit method (SDIRK 4(3), settings.integration_tool = 'sdirk4') on a shipped plant against scipy's Radau at 1e-11"""
    ex = CASES[name]
    sim = make_simulator(name, hostemu, integration_tool="sdirk4")
    m = sim.model
    p = plant.p_vector(m, plant.PLANT_P[name])
    x = ex.X0.copy()
    for k in range(2):
        u = np.array(U_TEST[name]) * (1.0 + 0.1 * k)
        r = sim.make_step_batch(x[None, :], u)
        assert r["status"][0] == 2, r["status"]                     # bit 1: implicit method, bit 0 clear
        ref = plant.plant_step(m, x, u, p, T_STEP[name])
        assert np.max(np.abs(r["x"][0] - ref) / np.maximum(1.0, np.abs(ref))) < tol, (name, k, r["x"][0], ref)
        x = ref


def stiff_model(mu=1e4):
    """Van der Pol oscillator in Lienard coordinates with stiffness parameter mu - no shipped example is stiff"""
    from do_mpc_amd import Model
    m = Model("continuous")
    x = m.set_variable("_x", "x", (2, 1))
    u = m.set_variable("_u", "u")
    m.set_rhs("x", np.array([[0.0], [0.0]]) + _vdp(x, u, mu))
    m.setup()
    return m


def _vdp(x, u, mu):
    from do_mpc_amd.sym import vertcat
    return vertcat(x[1], mu * ((1.0 - x[0] * x[0]) * x[1]) - x[0] + u)


def check_stiff_plant(hostemu):
    """A stiff plant (Van der Pol, mu = 1e4, two time units: the explicit pair would need ~2e4 steps): the explicit pair alone runs into its step limit (status 1, as
    before); the default ('cvodes': explicit first) notices that, repeats the interval with the implicit method and agrees with
    scipy's Radau; 'sdirk4' gives the same result directly.  The reference integrates every plant with CVODES / IDAS."""
    from scipy.integrate import solve_ivp
    m = stiff_model()
    x0, u0, T = np.array([2.0, 0.0]), np.array([0.1]), 2.0

    def make(tool, **kw):
        sim = Simulator(m)
        sim.set_param(t_step=T, abstol=1e-10, reltol=1e-10, integration_tool=tool, **kw)
        if hostemu:
            hdr = sim._lower()
            h = hdr.rsplit('PLANT_MODEL_HASH "', 1)[1].split('"')[0]
            sim.setup(_lib_path=hostemu_build.plant_hostemu_library(hdr, h, OUT), _code_object="")
        else:
            sim.setup()
        return sim

    f = lambda t, y: np.array([y[1], 1e4 * (1.0 - y[0] ** 2) * y[1] - y[0] + u0[0]])      # noqa: E731
    ref = solve_ivp(f, (0.0, T), x0, method="Radau", rtol=1e-12, atol=1e-13).y[:, -1]
    r = make("dopri5", max_steps=3000).make_step_batch(x0[None, :], u0)
    assert r["status"][0] == 1 and r["n_steps"][0] == 3000                 # explicit only: stiff -> step limit
    r = make("cvodes", integration_opts={"explicit_limit": 1500}).make_step_batch(x0[None, :], u0)
    assert r["status"][0] == 2                                             # implicit repeat, finished
    assert np.max(np.abs(r["x"][0] - ref) / np.maximum(1e-3, np.abs(ref))) < 1e-6, (r["x"][0], ref)
    assert r["n_steps"][0] < 1500 + 1200                                   # (the implicit method needs a few hundred steps)
    r2 = make("sdirk4").make_step_batch(x0[None, :], u0)
    assert r2["status"][0] == 2 and np.max(np.abs(r2["x"][0] - r["x"][0])) < 1e-12
    # a batch in which only some samples are stiff: every sample picks its own method
    X = np.array([[2.0, 0.0], [0.0, 0.0]])
    rb = make("cvodes", integration_opts={"explicit_limit": 1500}).make_step_batch(X, np.array([0.1]))
    assert rb["status"][0] == 2 and rb["status"][1] in (0, 2)
    # make_step (the closed-loop entry point) accepts a sample the implicit method finished (status bit 1 is not a failure) ...
    for tool, kw in (("sdirk4", {}), ("cvodes", {"integration_opts": {"explicit_limit": 1500}})):
        sim = make(tool, **kw)
        sim.x0 = x0
        y = sim.make_step(u0.reshape(-1, 1))
        assert np.max(np.abs(np.ravel(y) - ref) / np.maximum(1e-3, np.abs(ref))) < 1e-6, (tool, y, ref)
    # ... and still raises when the integration did not reach t_step
    sim = make("dopri5", max_steps=3000)
    sim.x0 = x0
    try:
        sim.make_step(u0.reshape(-1, 1))
    except RuntimeError as e:
        assert "did not reach t_step" in str(e)
    else:
        raise AssertionError("make_step accepted a sample that ran into the step limit")


def check_z_start_does_not_depend_on_call_history(hostemu):
    """the Newton iteration for the algebraic states of a DAE plant starts from `simulator.z0` in every make_step_batch call -
    a batch of unrelated samples must not depend on which samples occupied its rows in an earlier call.  Double inverted pendulum: two
    samples, then the same two in the other order: the rows of the second call are bit for bit the permuted rows of the first.
    `carry_z=True` (rows = the same trajectories, what make_step uses) continues from the previous values instead."""
    name = "dip"
    ex = CASES[name]
    sim = make_simulator(name, hostemu)
    X = np.vstack([ex.X0, ex.X0 * 0.5 + 0.1])
    u = np.array(U_TEST[name])
    a = sim.make_step_batch(X, u)
    b = sim.make_step_batch(X[::-1].copy(), u)
    assert a["status"].tolist() == [0, 0] and np.array_equal(b["x"], a["x"][::-1])
    c = sim.make_step_batch(X[::-1].copy(), u, carry_z=True)       # rows continue from the values row 0 / 1 found last: same roots
    assert np.max(np.abs(c["x"] - b["x"])) < 1e-9
