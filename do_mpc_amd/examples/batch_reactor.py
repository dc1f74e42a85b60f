"""Fed-batch bioreactor, economic NMPC.

Equations / tuning: /root/reference/examples/batch_reactor/template_model.py:34-74,
template_mpc.py:34-88, initial state main.py:56-61.
"""
import numpy as np

from .. import MPC, Model

MU_M, K_M, K_I, V_PAR, Y_P = 0.02, 0.05, 5.0, 0.004, 1.2


def build_model(symvar_type="SX"):
    mdl = Model("continuous", symvar_type)
    X = mdl.set_variable("_x", "X_s")
    S = mdl.set_variable("_x", "S_s")
    P = mdl.set_variable("_x", "P_s")
    V = mdl.set_variable("_x", "V_s")
    feed = mdl.set_variable("_u", "inp")
    Yx = mdl.set_variable("_p", "Y_x")
    Sin = mdl.set_variable("_p", "S_in")
    growth = MU_M * S / (K_M + S + (S ** 2 / K_I))
    mdl.set_rhs("X_s", growth * X - feed / V * X)
    mdl.set_rhs("S_s", -growth * X / Yx - V_PAR * X / Y_P + feed / V * (Sin - S))
    mdl.set_rhs("P_s", V_PAR * X - feed / V * P)
    mdl.set_rhs("V_s", feed)
    mdl.setup()
    return mdl


def build_mpc(model, silence_solver=True, n_horizon=20, **overrides):
    mpc = MPC(model)
    st = mpc.settings
    st.n_horizon, st.n_robust, st.open_loop = n_horizon, 0, 0
    st.t_step = 1.0
    st.state_discretization, st.collocation_type = "collocation", "radau"
    st.collocation_deg, st.collocation_ni = 2, 2
    st.store_full_solution = True
    for k, v in overrides.items():
        setattr(st, k, v)
    if silence_solver:
        st.supress_ipopt_output()
    mpc.set_objective(mterm=-model.x["P_s"], lterm=-model.x["P_s"])
    mpc.set_rterm(inp=1.0)
    for k, v in dict(X_s=0.0, S_s=-0.01, P_s=0.0, V_s=0.0).items():
        mpc.bounds["lower", "_x", k] = v
    mpc.bounds["upper", "_x", "X_s"] = 3.7
    mpc.bounds["upper", "_x", "P_s"] = 3.0
    mpc.bounds["lower", "_u", "inp"] = 0.0
    mpc.bounds["upper", "_u", "inp"] = 0.2
    mpc.set_uncertainty_values(Y_x=np.array([0.5, 0.4, 0.3]), S_in=np.array([200.0, 220.0, 180.0]))
    mpc.setup()
    return mpc


X0 = np.array([1.0, 0.5, 0.0, 120.0])
