"""Shared helpers of the plant-integrator tests (host emulation on CPU, HIP path under -m gpu)."""
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
        sim.setup(_lib_path=build.plant_hostemu_library(hdr, h, OUT), _code_object="")
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
