"""Moving horizon estimation: the restated reference NLP against IPOPT's stored run, the oracle's
interior-point method on it, and the product (host emulation of the kernels; the HIP twin is in test_gpu_parity.py)."""
import numpy as np
import pytest

import hostemu
import parity_common as pc
from do_mpc_amd.examples import rotating_masses as ex
from oracle import ipm


def test_golden_points_are_kkt_points_of_the_restated_mhe_nlp():
    """oracle/mhe.py restates _mhe.py:1030-1211 in the reference's own variable / row order: IPOPT's stored solutions
    (results_rotatingMasses.pkl, estimator record: opt_x 423, rows 450 - incl. the nl_cons rows of the last collocation point
    that the reference appends twice) satisfy its constraints and, with the stored multipliers, its stationarity conditions"""
    nlp = pc.oracle_mhe()
    g = pc.golden("rotating_masses")
    OX, OP, LG = g["estimator._opt_x_num"], g["estimator.opt_p_num"], g["estimator._lam_g_num"]
    assert (nlp.n_opt_x, nlp.n_opt_p, nlp.n_g) == (OX.shape[1], OP.shape[1], LG.shape[1]) == (423, 322, 450)
    for k in range(5):
        x, p, lam = OX[k], OP[k], LG[k]
        gv = nlp.g(x, p)
        eq = nlp.lbg == nlp.ubg
        assert np.max(np.abs(gv[eq])) < 1e-11 and np.max(gv[~eq] - nlp.ubg[~eq]) < 0.0
        rd = nlp.grad(x, p) + nlp.jac(x, p).T @ lam
        inside = (x > nlp.lbx + 1e-6) & (x < nlp.ubx - 1e-6)
        assert np.max(np.abs(rd[inside])) < 1e-9          # (measured 3e-11 ... 3e-10)


def test_oracle_ipm_reproduces_the_stored_mhe_solutions():
    """the oracle's interior-point method on the restated NLP, warm-started like the reference (previous solution as initial guess):
    steps 0, 1, 4 to the last bit (3e-16: IPOPT's iterates), step 2 to 8e-9; step 3 (45 iterations, inertia corrections in most of
    them) to 3e-6 - there the curvature test of the oracle and IPOPT's inertia count part ways"""
    nlp = pc.oracle_mhe()
    g = pc.golden("rotating_masses")
    OX, OP = g["estimator._opt_x_num"], g["estimator.opt_p_num"]
    xi = nlp.initial_guess(np.zeros(8), np.zeros(2), 1e-4)
    tol = [1e-12, 1e-12, 1e-7, 1e-5, 1e-12]
    for k in range(5):
        r = ipm.solve(nlp, xi, OP[k])
        assert r["stats"]["success"]
        assert pc.relerr(r["x"], OX[k]) < tol[k], (k, pc.relerr(r["x"], OX[k]))
        xi = r["x"]


def make_mhe(**kw):
    with hostemu.patched():
        return ex.build_mhe(ex.build_model(), **kw)


def test_mhe_golden_replay():
    pc.check_mhe_golden_replay(make_mhe)


def test_mhe_surface_and_refusals():
    from do_mpc_amd.estimator import MHE
    from do_mpc_amd.examples import CASES
    m = ex.build_model()
    mhe = MHE(m, ["Theta_1"])
    assert mhe._p_est.names == ["Theta_1"] and mhe._p_set.names == ["P_p", "Theta_2", "Theta_3"]
    with pytest.raises(Exception, match="solely depending"):
        mhe.set_objective(m.x["phi_1"] ** 2, m.x["phi_1"] ** 2)          # stage cost: w, v, tvp, p only (_mhe.py:585-589)
    with pytest.raises(AssertionError):
        MHE(m, ["not_a_parameter"])
    with pytest.raises(NotImplementedError, match="noise"):          # (a measurement without its own noise term)
        e = MHE(CASES["oscillating_masses"].build_model())
        e.settings.n_horizon, e.settings.t_step = 4, 0.5
        e.set_default_objective(np.eye(4))
        e.setup()


def test_mpc_plant_mhe_closed_loop_reproduces_the_reference_run():
    """The reference's test of the example as a whole (testing/test_rotating_oscillating_masses_mhe_mpc.py:77-110): controller ->
    plant -> estimator -> controller, 5 steps, true initial state random (seed 99), both optimisers started from 0.  Controller
    and estimator on the product kernels (host emulation), the plant = tests/plant.py (scipy Radau for CVODES); inputs, plant
    states and estimates against the stored run at the reference's own tolerance (1e-8; measured 2e-11 / 2e-11 / 4e-11)."""
    import plant
    g = pc.golden("rotating_masses")
    model = ex.build_model()
    with hostemu.patched():
        mpc = ex.build_mpc(model)
        mhe = ex.build_mhe(model)
    rng = np.random.RandomState(99)
    x_true = rng.rand(model.n_x) - 0.5
    x_est = np.zeros(model.n_x)
    mpc.x0 = x_est
    mhe.x0 = x_est
    mhe.p_est0 = 1e-4
    mpc.set_initial_guess()
    mhe.set_initial_guess()
    p_true = plant.p_vector(model, {"P_p": 0.0, "Theta_1": 2.25e-4, "Theta_2": 2.25e-4, "Theta_3": 2.25e-4})
    worst = np.zeros(3)
    for k in range(5):
        assert pc.relerr(x_true, g["simulator._x"][k]) < 1e-8
        u0 = mpc.make_step(x_est).ravel()
        x_true = plant.plant_step(model, x_true, u0, p_true, 0.1)
        y = np.asarray(model._meas_fun.eval(x_true, u0, np.zeros(0), np.zeros(model.n_tvp), p_true, np.zeros(model.n_v))[0]).ravel()
        x_est = mhe.make_step(y).ravel()
        worst = np.maximum(worst, [pc.relerr(u0, g["mpc._u"][k]), pc.relerr(y, g["estimator._y"][k]), pc.relerr(mhe.data["_x"][k], g["estimator._x"][k])])
    assert np.all(worst < 1e-8), worst


def test_mhe_batch_of_estimation_problems():
    pc.check_mhe_batch(make_mhe)


def test_mhe_with_process_noise_against_the_oracle():
    def make():
        with hostemu.patched():
            return ex.build_mhe_w(ex.build_model(process_noise=True))
    pc.check_mhe_with_process_noise(make)


def test_mhe_inputs_measured_without_noise_against_the_oracle():
    def make():
        with hostemu.patched():
            return ex.build_mhe_w(ex.build_model(process_noise=True, input_meas_noise=False), max_batch=2)
    pc.check_mhe_inputs_measured_without_noise(make)


@pytest.mark.parametrize("single_slack", [False, True], ids=["slack_per_stage", "single_slack"])
def test_mhe_soft_constraint_against_the_oracle(single_slack):
    def make(**kw):
        with hostemu.patched():
            return ex.build_mhe_w(ex.build_model(process_noise=True), **kw)
    pc.check_mhe_soft_constraint(make, single_slack)


def test_mhe_scaling_of_states_inputs_and_estimated_parameters():
    def make(**kw):
        with hostemu.patched():
            return ex.build_mhe_w(ex.build_model(process_noise=True), max_batch=1, **kw)
    pc.check_mhe_scaling_invariance(make)


def test_mhe_for_a_model_with_algebraic_states():
    def make(dae):
        with hostemu.patched():
            return ex.build_mhe_w(ex.build_model(process_noise=True, dae=dae))
    pc.check_mhe_dae_equals_ode(make)


def test_mhe_make_step_for_a_model_with_algebraic_states():
    def make(dae):
        with hostemu.patched():
            return ex.build_mhe(ex.build_model(dae=dae))
    pc.check_mhe_dae_make_step(make)


def test_discrete_time_mhe_against_the_oracle():
    from do_mpc_amd.examples import oscillating_masses as om

    def make():
        with hostemu.patched():
            return om.build_mhe(om.build_model(estimation=True))
    pc.check_discrete_mhe(make)


def test_discrete_time_mhe_for_a_model_with_algebraic_states():
    from do_mpc_amd.examples import oscillating_masses as om

    def make(dae):
        with hostemu.patched():
            return om.build_mhe(om.build_model(estimation=True, dae=dae))
    pc.check_discrete_mhe_dae_equals_ode(make)


def test_watchdog_on_the_straggler_of_the_cold_estimator_batch():
    pc.check_watchdog_on_mhe_straggler(make_mhe)

