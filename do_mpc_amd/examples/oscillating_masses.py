"""Two oscillating masses, discrete-time linear MPC (BASELINE configs[0], the CPU-plumbing case).

Equations / tuning: /root/reference/examples/oscillating_masses_discrete/template_model.py:34-73,
template_mpc.py:34-74; test initial state /root/reference/testing/test_oscillating_masses_discrete.py:82-84.
"""
import numpy as np

from .. import MPC, Model
from ..sym import sum1

A_D = np.array([[0.763, 0.460, 0.115, 0.020],
                [-0.899, 0.763, 0.420, 0.115],
                [0.115, 0.020, 0.763, 0.460],
                [0.420, 0.115, -0.899, 0.763]])
B_D = np.array([[0.014], [0.063], [0.221], [0.367]])


def build_model(symvar_type="SX", estimation=False, dae=False):
    """estimation: the variant for the discrete-time estimator build_mhe (not in the reference's example): the two positions are
    measured with noise, process noise on all four states.
    dae (with estimation): the same plant with the free response `ax = A x` as algebraic states - an equivalent model for the
    estimator's discrete-time DAE path"""
    mdl = Model("discrete", symvar_type)
    x = mdl.set_variable(var_type="_x", var_name="x", shape=(4, 1))
    u = mdl.set_variable(var_type="_u", var_name="u", shape=(1, 1))
    mdl.set_expression(expr_name="cost", expr=sum1(x ** 2))
    if estimation:
        from ..sym import vertcat
        mdl.set_meas("pos_meas", vertcat(x[0], x[2]))
    if dae:
        ax = mdl.set_variable(var_type="_z", var_name="ax", shape=(4, 1))
        mdl.set_alg("ax", ax - A_D @ x)
        mdl.set_rhs("x", ax + B_D @ u, process_noise=estimation)
    else:
        mdl.set_rhs("x", A_D @ x + B_D @ u, process_noise=estimation)
    mdl.setup()
    return mdl


def build_mpc(model, silence_solver=True, n_horizon=7, custom_rterm=None, **overrides):
    mpc = MPC(model)
    st = mpc.settings
    st.n_robust, st.n_horizon, st.t_step = 0, n_horizon, 0.5
    st.store_full_solution = True
    for k, v in overrides.items():
        setattr(st, k, v)
    if silence_solver:
        st.supress_ipopt_output()
    mpc.set_objective(mterm=model.aux["cost"], lterm=model.aux["cost"])
    if custom_rterm is not None:          # user-defined input penalty: expression in model.x / model.u / mpc.u_prev (_mpc.py:593-677)
        mpc.set_rterm(rterm=(RTERM_VARIANTS[custom_rterm] if isinstance(custom_rterm, str) else custom_rterm)(model, mpc))
    else:
        mpc.set_rterm(u=1e-4)
    limit = np.array([[4.0], [10.0], [4.0], [10.0]])
    mpc.bounds["lower", "_x", "x"] = -limit
    mpc.bounds["upper", "_x", "x"] = limit
    mpc.bounds["lower", "_u", "u"] = -0.5
    mpc.bounds["upper", "_u", "u"] = 0.5
    mpc.setup()
    return mpc


def _x0():
    rng = np.random.RandomState(99)
    return rng.rand(4) - 0.5


X0 = _x0()


def _rterm_custom(model, mpc):
    du = model.u["u"] - mpc.u_prev["u"]
    return 1e-2 * du ** 2 + 1e-1 * du ** 4 + 1e-2 * model.x["x", 0] ** 2 * du ** 2


RTERM_VARIANTS = {"custom": _rterm_custom}


def build_mhe(model, silence_solver=True, **overrides):
    """A discrete-time moving horizon estimator on build_model(estimation=True): horizon 8, weights as numbers, one nl_cons row.
    No stored run exists for it: compared with the oracle's solve of the restated NLP (oracle/mhe.py)."""
    from ..estimator import MHE
    mhe = MHE(model, [])
    st = mhe.settings
    st.n_horizon, st.t_step, st.store_full_solution = 8, 0.5, True
    for k, v in overrides.items():
        setattr(st, k, v)
    if silence_solver:
        st.supress_ipopt_output()
    mhe.set_default_objective(0.5 * np.eye(4), 10.0 * np.eye(2), None, 5.0 * np.eye(4))
    limit = np.array([[4.0], [10.0], [4.0], [10.0]])
    mhe.bounds["lower", "_x", "x"] = -limit
    mhe.bounds["upper", "_x", "x"] = limit
    mhe.bounds["lower", "_u", "u"] = -0.5
    mhe.bounds["upper", "_u", "u"] = 0.5
    mhe.set_nl_cons("x1_ub", model.x["x", 1] - 3.0, 0)
    mhe.setup()
    return mhe
