"""The reference's shipped template_model.py / template_mpc.py run UN-EDITED on this backend.

Needs /root/reference (present in the build container, absent on the GPU box -> skipped there).
The files are imported from where they lie; nothing is copied.  Each template must give the same structure, bounds and
scalings as the in-repo restatement in do_mpc_amd/examples/ AND THE SAME GENERATED HEADER, character by character (the
restatements associate their operations like the templates; the expression DAG is hash-consed, so equal expressions give
equal text): the model hash - the name of the gfx950 code object - is the same, i.e. every `-m gpu` parity test of an
in-repo case (tests/test_gpu_parity.py: golden replays, Newton directions, oracle solves) runs EXACTLY the code object that
the un-edited template lowers to.  The hashes are pinned in tests/golden/template_hashes.json, which the GPU box checks
against the code objects it runs (tests/test_gpu_parity.py::test_code_objects_are_the_ones_the_unedited_templates_lower_to).
On the host emulation the un-edited templates also reproduce the golden u0 / closed loops directly.
"""
import hostemu_build
import importlib.util
import os
import sys

import numpy as np
import pytest

import hostemu
from do_mpc_amd import casadi_compat
from do_mpc_amd.examples import CASES

REF = "/root/reference/examples"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not available")

DIRS = {"industrial_poly": "industrial_poly", "CSTR": "CSTR", "batch_reactor": "batch_reactor",
        "oscillating_masses": "oscillating_masses_discrete", "kinematic_bicycle": "kinematic_bicycle_model",
        "dynamic_bicycle": "dynamic_bicycle_model", "kite": "kite",
        "rotating_masses": "rotating_oscillating_masses_mhe_mpc",
        "oscillating_masses_dae": "oscillating_masses_discrete_dae", "dip": "double_inverted_pendulum"}       # DAE models (`_z`)
MPC_ARGS = {"kite": (10.0, 6.0)}           # template_mpc(model, w_ref, E_0, h_min=100): main.py draws them at random
MODEL_ARGS = {"dip": ([{"x": 0., "y": 0.6, "r": 0.3}],)}     # template_model(obstacles): the obstacle of testing/test_DIP.py:70-73
HASHES = os.path.join(os.path.dirname(__file__), "golden", "template_hashes.json")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture()
def compat():
    names = casadi_compat.install()
    yield
    casadi_compat.uninstall(names)


@pytest.mark.parametrize("name", list(DIRS))
def test_unedited_templates_lower_to_the_same_model(name, compat):
    d = os.path.join(REF, DIRS[name])
    tm = _load(os.path.join(d, "template_model.py"), f"ref_{name}_template_model")
    tc = _load(os.path.join(d, "template_mpc.py"), f"ref_{name}_template_mpc")
    with hostemu.patched():
        ref_model = tm.template_model(*MODEL_ARGS.get(name, ()))
        ref_mpc = tc.template_mpc(ref_model, *MPC_ARGS.get(name, ()), silence_solver=True)
        ours = CASES[name].build_mpc(CASES[name].build_model())
    # the un-edited template and the in-repo restatement lower to the same text -> the same gfx950 code object
    assert ref_mpc.generated_header == ours.generated_header
    import json
    assert json.load(open(HASHES))[name] == ref_mpc.model_hash, "re-run tools/template_hashes.py"
    assert ref_mpc.structure.n_opt_x == ours.structure.n_opt_x
    assert ref_mpc.structure.n_g == ours.structure.n_g
    assert np.array_equal(ref_mpc._lb_opt_x.master, ours._lb_opt_x.master)
    assert np.array_equal(ref_mpc._ub_opt_x.master, ours._ub_opt_x.master)
    assert np.array_equal(ref_mpc.opt_x_scaling.master, ours.opt_x_scaling.master)
    # same model functions (the in-repo restatement factors the expressions differently, so the
    # generated text may differ; the values must not)
    rng = np.random.default_rng(0)
    m1, m2 = ref_model, ours.model
    for _ in range(5):
        x = ours._x0.master * 0 + CASES[name].X0 * (1 + 0.01 * rng.standard_normal(m1.n_x)) + 0.01 * rng.standard_normal(m1.n_x)
        u = 0.5 * (ours._u_lb.master + ours._u_ub.master) * (1 + 0.01 * rng.standard_normal(m1.n_u))
        p = ours.p_fun(0.0).master[:m1.n_p]
        z = 0.1 * rng.standard_normal(m1.n_z)
        args = (x, u, z, np.zeros(m1.n_tvp), p, np.zeros(m1.n_w))
        r1 = m1._rhs_fun.eval(*args)[0]
        r2 = m2._rhs_fun.eval(*args)[0]
        assert np.allclose(r1, r2, rtol=1e-12, atol=1e-12)
    assert "DOMPC_NX %d" % m1.n_x in ref_mpc.generated_header


@pytest.mark.parametrize("name", ["industrial_poly", "CSTR", "rotating_masses"])
def test_unedited_templates_with_mx_symbols_lower_to_the_same_model(name, compat):
    """the reference's tests build every example twice, `template_model('SX')` and `template_model('MX')`
    (testing/test_industrial_poly.py:57-63): both symbol types are the same scalar expression graph here - same generated header"""
    import json
    d = os.path.join(REF, DIRS[name])
    tm = _load(os.path.join(d, "template_model.py"), f"ref_{name}_template_model_mx")
    tc = _load(os.path.join(d, "template_mpc.py"), f"ref_{name}_template_mpc_mx")
    with hostemu.patched():
        mpc = tc.template_mpc(tm.template_model("MX"), silence_solver=True)
    assert json.load(open(HASHES))[name] == mpc.model_hash


def test_unedited_mhe_template_lowers_to_the_same_estimator_and_replays_the_stored_run(compat):
    """template_mhe.py of examples/rotating_oscillating_masses_mhe_mpc, un-edited (`do_mpc.estimator.MHE`, `mhe._p_est[...]`,
    `mhe.data._y`-driven y_fun, default objective with symbolic weights): the chain problem it lowers to has the same generated
    header as the in-repo restatement (do_mpc_amd/examples/rotating_masses.py:build_mhe) - the code object of the GPU parity
    test - and, on the host emulation, reproduces the estimator record of the reference's test run"""
    import parity_common as pc
    d = os.path.join(REF, DIRS["rotating_masses"])
    tm = _load(os.path.join(d, "template_model.py"), "ref_rot_template_model_mhe")
    te = _load(os.path.join(d, "template_mhe.py"), "ref_rot_template_mhe")
    ex = CASES["rotating_masses"]

    def make():
        with hostemu.patched():
            return te.template_mhe(tm.template_model(), silence_solver=True)
    with hostemu.patched():
        ours = ex.build_mhe(ex.build_model())
    mhe = pc.check_mhe_golden_replay(make)
    assert mhe._mpc.generated_header == ours._mpc.generated_header
    import json
    assert json.load(open(HASHES))["rotating_masses_mhe"] == mhe._mpc.model_hash, "re-run tools/template_hashes.py"


def test_unedited_industrial_poly_template_reproduces_golden_first_step(compat):
    d = os.path.join(REF, "industrial_poly")
    tm = _load(os.path.join(d, "template_model.py"), "ref_ip_template_model2")
    tc = _load(os.path.join(d, "template_mpc.py"), "ref_ip_template_mpc2")
    with hostemu.patched():
        mpc = tc.template_mpc(tm.template_model(), silence_solver=True)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "industrial_poly.npz"))
    # main.py:59-73 sets x0 through the simulator's struct; same numbers here
    mpc.x0 = g["mpc._x"][0]
    mpc.set_initial_guess()
    u0 = mpc.make_step(g["mpc._x"][0]).ravel()
    assert np.max(np.abs(u0 - g["mpc._u"][0]) / np.maximum(1, np.abs(g["mpc._u"][0]))) < 1e-6


@pytest.mark.parametrize("name", ["CSTR", "industrial_poly"])
def test_unedited_template_simulator_closes_the_loop_like_main_py(name, compat):
    """template_model.py + template_mpc.py + template_simulator.py of the reference, un-edited, in the loop of
    examples/*/main.py:105-108 (mpc.make_step -> simulator.make_step -> estimator.make_step) - controller and plant on the
    host emulation of the kernels here - against the golden trajectory of the reference's own test."""
    import simulator_common as sc
    from do_mpc_amd import build
    import do_mpc                                  # the stand-in registered by the fixture
    d = os.path.join(REF, DIRS[name])
    tm = _load(os.path.join(d, "template_model.py"), f"ref_{name}_tm3")
    tc = _load(os.path.join(d, "template_mpc.py"), f"ref_{name}_tc3")
    tsim = _load(os.path.join(d, "template_simulator.py"), f"ref_{name}_ts3")
    model = tm.template_model()
    orig_setup = do_mpc.simulator.Simulator.setup

    def setup_on_hostemu(self):                    # (no GPU here: the plant kernel's host emulation; same entry point)
        hdr = self._lower()
        h = hdr.rsplit('PLANT_MODEL_HASH "', 1)[1].split('"')[0]
        orig_setup(self, _lib_path=hostemu_build.plant_hostemu_library(hdr, h, sc.OUT), _code_object="")
    do_mpc.simulator.Simulator.setup = setup_on_hostemu
    try:
        with hostemu.patched():
            mpc = tc.template_mpc(model, silence_solver=True)
        simulator = tsim.template_simulator(model)
    finally:
        do_mpc.simulator.Simulator.setup = orig_setup
    estimator = do_mpc.estimator.StateFeedback(model)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"{name}.npz"))
    x0 = g["mpc._x"][0].reshape(-1, 1)
    mpc.x0 = x0
    simulator.x0 = x0
    estimator.x0 = x0
    mpc.set_initial_guess()
    for k in range(3):
        u0 = mpc.make_step(x0)
        y_next = simulator.make_step(u0)
        x0 = estimator.make_step(y_next)
        assert np.max(np.abs(u0.ravel() - g["mpc._u"][k]) / np.maximum(1, np.abs(g["mpc._u"][k]))) < 2e-6
        assert np.max(np.abs(x0.ravel() - g["mpc._x"][k + 1]) / np.maximum(1, np.abs(g["mpc._x"][k + 1]))) < 2e-6


def test_unedited_batch_reactor_differentiator_templates_and_sensitivity_query(compat):
    """examples/batch_reactor_differentiator: its template_model / template_mpc un-edited, then the calls its main.py makes
    (main.py:131-168: DoMPCDifferentiator(mpc), settings, differentiate(), sens_num[...] with casadi.tools.indexf)."""
    import do_mpc
    from casadi.tools import indexf
    d = os.path.join(REF, "batch_reactor_differentiator")
    tm = _load(os.path.join(d, "template_model.py"), "ref_brd_tm")
    tc = _load(os.path.join(d, "template_mpc.py"), "ref_brd_tc")
    with hostemu.patched():
        model = tm.template_model()
        mpc = tc.template_mpc(model)
    x0 = np.array([1.0, 0.5, 0.0, 120.0]).reshape(-1, 1)
    mpc.x0 = x0
    mpc.set_initial_guess()
    mpc.make_step(x0)
    nlp_diff = do_mpc.differentiator.DoMPCDifferentiator(mpc)
    nlp_diff.settings.check_LICQ = False
    nlp_diff.settings.check_rank = False
    nlp_diff.settings.lin_solver = 'scipy'
    nlp_diff.differentiate()
    du0dx0_num = nlp_diff.sens_num["dxdp", indexf["_u", 0, 0], indexf["_x0"]]
    du0du_prev_num = nlp_diff.sens_num["dxdp", indexf["_u", 0, 0], indexf["_u_prev"]].full()
    assert du0dx0_num.shape == (model.n_u, model.n_x) and du0du_prev_num.shape == (model.n_u, model.n_u)
    assert np.all(np.isfinite(du0dx0_num))


def test_unedited_sampling_tool_chain_script_reproduces_its_golden_table(compat, tmp_path, monkeypatch):
    """examples/tools/sampling/regular/test_fun/sampling_test.py un-edited (`do_mpc.sampling.SamplingPlanner / Sampler /
    DataHandler`), compared for equality with the reference's stored table like testing/test_sampling_tools.py:55-67."""
    import pickle
    mod = _load(os.path.join(REF, "tools", "sampling", "regular", "test_fun", "sampling_test.py"), "ref_sampling_test")
    monkeypatch.chdir(tmp_path)                          # the script writes ./sample_results/
    res, res1, res2 = mod.main()
    with open("/root/reference/testing/results/res_sampling_test_test_fun.pkl", "rb") as f:
        ref = pickle.load(f)
    assert res == ref["res"] and res1 == ref["res1"] and res2 == ref["res2"]


def test_unedited_triple_tank_model_and_simulator_reproduce_the_golden_plant_trajectory(compat):
    """examples/triple_tank_ekf: template_model.py (discrete model with sign / sqrt / fabs, one `_p`, one `_tvp`, a
    measurement) and template_simulator.py (p_fun, a tvp_fun that switches at t = 50) un-edited on the batched plant kernel's
    host emulation, driven like testing/test_triple_tank_EKF.py:88-101 (200 steps, constant input, seeded measurement
    noise) - states and noisy measurements against results_triple_tank_ekf.pkl."""
    import simulator_common as sc
    from do_mpc_amd import build
    import do_mpc
    d = os.path.join(REF, "triple_tank_ekf")
    tm = _load(os.path.join(d, "template_model.py"), "ref_tt_tm")
    tsim = _load(os.path.join(d, "template_simulator.py"), "ref_tt_ts")
    model = tm.template_model()
    orig_setup = do_mpc.simulator.Simulator.setup

    def setup_on_hostemu(self):
        hdr = self._lower()
        h = hdr.rsplit('PLANT_MODEL_HASH "', 1)[1].split('"')[0]
        orig_setup(self, _lib_path=hostemu_build.plant_hostemu_library(hdr, h, sc.OUT), _code_object="")
    do_mpc.simulator.Simulator.setup = setup_on_hostemu
    try:
        simulator = tsim.template_simulator(model)
    finally:
        do_mpc.simulator.Simulator.setup = orig_setup
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "triple_tank.npz"))
    simulator.x0 = np.array([2, 2.8, 2.7]).reshape([-1, 1])
    simulator.set_initial_guess()
    np.random.seed(42)
    ys = []
    for k in range(200):
        u0 = np.array([0.0001, 0.0001]).reshape([-1, 1])
        ys.append(np.asarray(simulator.make_step(u0, v0=0.001 * np.random.randn(model.n_v, 1))).ravel())
    X, Y = simulator.data["_x"], np.array(ys)
    assert X.shape == g["simulator._x"].shape
    assert np.max(np.abs(X - g["simulator._x"])) < 1e-8          # (the tolerance of the reference's own test)
    assert np.max(np.abs(Y - g["simulator._y"])) < 1e-8
    assert np.array_equal(simulator.data["_tvp"], g["simulator._tvp"]) and np.array_equal(simulator.data["_p"], g["simulator._p"])


def test_unedited_approximate_mpc_controller_is_sampled_as_one_batch(compat):
    """examples/CSTR_approximate_mpc: its template_model / template_mpc un-edited (nominal CSTR, no nl_cons); the open-loop
    sampling of do_mpc.approximateMPC.AMPCSampler (_ampc_sampler.py:234-345: (x0, u_prev) uniform in the box of the bounds,
    one cold make_step per sample) as ONE batched solve = the per-sample loop."""
    from do_mpc_amd import sampling
    d = os.path.join(REF, "CSTR_approximate_mpc")
    tm = _load(os.path.join(d, "template_model.py"), "ref_ampc_tm")
    tc = _load(os.path.join(d, "template_mpc.py"), "ref_ampc_tc")
    model = tm.template_model()
    with hostemu.patched():
        one = tc.template_mpc(model, silence_solver=True)
        tc2 = _load(os.path.join(d, "template_mpc.py"), "ref_ampc_tc2")
        orig = tc2.do_mpc.controller.MPC.__init__

        def with_batch(self, *a, **k):                  # (the template does not know about max_batch; same object otherwise)
            orig(self, *a, **k)
            self.settings.max_batch = 4
        tc2.do_mpc.controller.MPC.__init__ = with_batch
        try:
            mpc = tc2.template_mpc(model, silence_solver=True)
        finally:
            tc2.do_mpc.controller.MPC.__init__ = orig
    lbx, ubx = one._x_lb.master, one._x_ub.master
    lbu, ubu = one._u_lb.master, one._u_ub.master
    assert np.all(np.isfinite(np.concatenate([lbx, ubx, lbu, ubu])))
    plan = sampling.sampling_plan_box(lbx, ubx, lbu, ubu, n_samples=4, seed=1)
    res = sampling.open_loop_samples(mpc, plan)
    for i in range(4):
        one.reset_history()
        one.x0 = plan["x0"][i]
        one.u0 = plan["u_prev"][i]
        one.set_initial_guess()
        u0 = one.make_step(plan["x0"][i]).ravel()
        assert bool(one.solver_stats["success"]) == bool(res["status"][i])
        if res["status"][i]:
            assert np.allclose(u0, res["u0"][i], rtol=1e-9, atol=1e-9)
            assert one.solver_stats["iter_count"] == res["iter_count"][i]
    assert res["status"].sum() >= 2
