"""Three rotating discs coupled by springs, driven by two stepper motors; NMPC tracking a time-varying set-point for
the middle disc (the controller of the reference's MHE + MPC example).

Equations / tuning: /root/reference/examples/rotating_oscillating_masses_mhe_mpc/template_model.py:34-99,
template_mpc.py:34-106, main.py:52-63.  The only shipped MPC example with time-varying parameters on the path: the
set-point `phi_2_set` is a random staircase (seed 999) read over the horizon; the model also declares the measurement
weights `P_v` (5x5 `_tvp`) and `P_p` (`_p`) that only its estimator uses - they are kept so that the parameter vector
has the reference's layout.  Two values of the inertia `Theta_1` are declared but `n_robust = 0`: one scenario, nominal.
"""
import numpy as np

from .. import MPC, Model
from ..sym import DM, vertcat, vertsplit

N_TRAJ = 400
X0 = np.zeros(8)                                       # main.py:58 (the controller starts from the estimator's x0 = 0)


def setpoint_trajectory():
    """template_mpc.py:63-73."""
    rng = np.random.RandomState(999)
    traj = [0.0]
    for _ in range(N_TRAJ):
        nxt = (0.5 - rng.rand()) * np.pi
        switch = rng.rand() >= 0.95
        traj.append((1 - switch) * traj[-1] + switch * nxt)
    return np.array(traj)


def build_model(symvar_type="SX", process_noise=False, dae=False, input_meas_noise=True):
    """process_noise: additive noise on the three angular accelerations (`set_rhs(..., process_noise=True)`; not in the reference's
    example - the estimator variant build_mhe_w uses it).
    dae: the same plant written with algebraic states - the twist `tw_i = phi_i - left_i` of the spring on the left of every disc
    as `_z` with its algebraic equation, read by the accelerations AND by the first measurement (`phi_1 = tw_1 + phi_m_1`): an
    equivalent model for the estimator's DAE path (same estimates as the ODE model; no stored run exists for it)
    input_meas_noise=False: the motor set-points are measured WITHOUT noise (`set_meas(..., meas_noise=False)`, what the reference's
    documentation suggests for measured inputs, _model.py:693-695, and its MHE example notebook does)"""
    mdl = Model("continuous", symvar_type)
    phi = vertcat(*[mdl.set_variable("_x", "phi_%d" % i) for i in (1, 2, 3)])
    dphi = mdl.set_variable("_x", "dphi", shape=(3, 1))
    phi_m_set = mdl.set_variable("_u", "phi_m_set", shape=(2, 1))
    phi_m = mdl.set_variable("_x", "phi_m", shape=(2, 1))
    mdl.set_variable("_tvp", "phi_2_set")
    mdl.set_variable("_p", "P_p")
    mdl.set_variable("_tvp", "P_v", shape=(5, 5))
    left = [phi_m[0], phi[0], phi[1]]
    right = [phi[1], phi[2], phi_m[1]]
    if dae:
        tw = mdl.set_variable("_z", "tw", shape=(3, 1))
        mdl.set_alg("twist", vertcat(*[tw[i] - (phi[i] - left[i]) for i in range(3)]))
        twist = [tw[i] for i in range(3)]
        mdl.set_meas("phi_1_meas", vertcat(tw[0] + phi_m[0], phi[1], phi[2]))
    else:
        twist = [phi[i] - left[i] for i in range(3)]
        mdl.set_meas("phi_1_meas", phi)
    mdl.set_meas("phi_m_set_meas", phi_m_set, meas_noise=input_meas_noise)
    th = [mdl.set_variable("_p", "Theta_%d" % i) for i in (1, 2, 3)]
    c = np.array([2.697, 2.66, 3.05, 2.86]) * 1e-3       # spring constants
    d = np.array([6.78, 8.01, 8.82]) * 1e-5              # friction
    for i in range(3):
        mdl.set_rhs("phi_%d" % (i + 1), dphi[i])
    mdl.set_rhs("dphi", vertcat(*[-c[i] / th[i] * twist[i] - c[i + 1] / th[i] * (phi[i] - right[i])
                                  - d[i] / th[i] * dphi[i] for i in range(3)]), process_noise=process_noise)
    mdl.set_rhs("phi_m", 1 / 1e-2 * (phi_m_set - phi_m))
    mdl.setup()
    return mdl


def build_mpc(model, silence_solver=True, **overrides):
    mpc = MPC(model)
    st = mpc.settings
    st.n_robust, st.n_horizon, st.t_step, st.store_full_solution = 0, 20, 0.1, True
    for k, v in overrides.items():
        setattr(st, k, v)
    if silence_solver:
        st.supress_ipopt_output()
    mpc.set_objective(mterm=DM(1), lterm=(model.x["phi_2"] - model.tvp["phi_2_set"]) ** 2)
    mpc.set_rterm(phi_m_set=1e-2)
    traj = setpoint_trajectory()
    tvp_template = mpc.get_tvp_template()

    def tvp_fun(t_now):
        ind = int(t_now / st.t_step)
        tvp_template["_tvp", :-1] = vertsplit(traj[ind:ind + st.n_horizon])   # (every entry of the stage, like the template)
        return tvp_template

    mpc.set_tvp_fun(tvp_fun)
    mpc.set_uncertainty_values(Theta_1=2.25e-4 * np.array([1.0, 1.1]), Theta_2=2.25e-4 * np.array([1.0]),
                               Theta_3=2.25e-4 * np.array([1.0]))
    mpc.bounds["lower", "_u", "phi_m_set"] = -5
    mpc.bounds["upper", "_u", "phi_m_set"] = 5
    mpc.setup()
    return mpc


def build_mhe(model, silence_solver=True, **overrides):
    """The example's estimator (/root/reference/examples/rotating_oscillating_masses_mhe_mpc/template_mhe.py:34-104): horizon 10,
    `Theta_1` estimated, default objective with P_x = 1e-4 I, the `_tvp` P_v = diag(1, 1, 1, 20, 20) and the parameter P_p = 1,
    bounds on the motor set-points and the angular velocities, the box of Theta_1 as two nl_cons rows checked at the
    collocation points; measurements from the data object."""
    from ..estimator import MHE
    mhe = MHE(model, ["Theta_1"])
    st = mhe.settings
    st.n_horizon, st.t_step, st.store_full_solution, st.nl_cons_check_colloc_points = 10, 0.1, True, True
    for k, v in overrides.items():
        setattr(st, k, v)
    if silence_solver:
        st.supress_ipopt_output()
    mhe.set_default_objective(1e-4 * np.eye(8), model.tvp["P_v"], model.p["P_p"])
    tvp_template = mhe.get_tvp_template()
    tvp_template["_tvp", :, "P_v"] = np.diag(np.array([1, 1, 1, 20, 20]))
    mhe.set_tvp_fun(lambda t_now: tvp_template)
    p_template = mhe.get_p_template()

    def p_fun(t_now):
        p_template["Theta_2"] = 2.25e-4
        p_template["Theta_3"] = 2.25e-4
        p_template["P_p"] = np.eye(1)
        return p_template

    mhe.set_p_fun(p_fun)
    y_template = mhe.get_y_template()

    def y_fun(t_now):
        n_steps = min(mhe.data["_y"].shape[0], st.n_horizon)
        for k in range(-n_steps, 0):
            y_template["y_meas", k] = mhe.data["_y"][k]
        return y_template

    mhe.set_y_fun(y_fun)
    mhe.bounds["lower", "_u", "phi_m_set"] = -5
    mhe.bounds["upper", "_u", "phi_m_set"] = 5
    mhe.bounds["lower", "_x", "dphi"] = -6
    mhe.bounds["upper", "_x", "dphi"] = 6
    mhe.set_nl_cons("p_est_lb", -mhe._p_est["Theta_1"] + 1e-5, 0)
    mhe.set_nl_cons("p_est_ub", mhe._p_est["Theta_1"] - 1e-3, 0)
    mhe.setup()
    return mhe


# a scaling of states, inputs and the estimated parameter for build_mhe_w (scaling-invariance tests)
MHE_W_SCALING = {("_x", "phi_1"): 2.0, ("_x", "phi_m"): 0.5, ("_x", "dphi"): 5.0, ("_u", "phi_m_set"): 3.0, ("_p_est", "Theta_1"): 1e-4}


def build_mhe_w(model, silence_solver=True, scaling=None, soft_limit=None, **overrides):
    """A second estimator on the model with process noise (build_model(process_noise=True)) for the paths the shipped example leaves
    out: `_w` as decision variables with weight P_w, numeric weights, the box of Theta_1 as bounds of `_p_est`, an nl_cons row on a
    state checked at the states only.  No stored run exists for it: compared with the oracle's solve of the restated NLP."""
    from ..estimator import MHE
    mhe = MHE(model, ["Theta_1"])
    st = mhe.settings
    st.n_horizon, st.t_step, st.store_full_solution, st.nl_cons_check_colloc_points = 6, 0.1, True, False
    for k, v in overrides.items():
        setattr(st, k, v)
    if silence_solver:
        st.supress_ipopt_output()
    mhe.set_default_objective(1e-4 * np.eye(8), np.diag([1.0, 1.0, 1.0, 20.0, 20.0][:model.n_v]), np.eye(1), 10.0 * np.eye(3))
    tvp_template = mhe.get_tvp_template()
    mhe.set_tvp_fun(lambda t_now: tvp_template)
    p_template = mhe.get_p_template()
    p_template["Theta_2"] = 2.25e-4
    p_template["Theta_3"] = 2.25e-4
    p_template["P_p"] = 1.0
    mhe.set_p_fun(lambda t_now: p_template)
    mhe.bounds["lower", "_u", "phi_m_set"] = -5
    mhe.bounds["upper", "_u", "phi_m_set"] = 5
    mhe.bounds["lower", "_x", "dphi"] = -6
    mhe.bounds["upper", "_x", "dphi"] = 6
    mhe.bounds["lower", "_p_est", "Theta_1"] = 1e-5
    mhe.bounds["upper", "_p_est", "Theta_1"] = 1e-3
    if soft_limit is None:
        mhe.set_nl_cons("phi_1_ub", model.x["phi_1"] - 1.5, 0)
    else:             # (limit, penalty): the row as a soft constraint (slack variables `_eps`: one per stage, or one with nl_cons_single_slack)
        mhe.set_nl_cons("phi_1_ub", model.x["phi_1"] - soft_limit[0], 0, soft_constraint=True, penalty_term_cons=soft_limit[1])
    for (group, name), v in (scaling or {}).items():       # e.g. {("_x", "dphi"): 5.0, ("_p_est", "Theta_1"): 1e-4}
        mhe.scaling[group, name] = v
    mhe.setup()
    return mhe
