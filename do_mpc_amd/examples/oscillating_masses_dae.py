"""Two oscillating masses as a discrete-time DAE: the successor state is an algebraic variable, x+ = z with
0 = z - A x - B u  (the reference's smallest model with `_z`).

Equations / tuning: /root/reference/examples/oscillating_masses_discrete_dae/template_model.py:34-75,
template_mpc.py:34-74; test initial state /root/reference/testing/test_oscillating_masses_discrete_dae.py.
"""
import numpy as np

from .. import MPC, Model
from ..sym import sum1
from .oscillating_masses import A_D, B_D, X0  # noqa: F401  (same plant, same test initial state)


def build_model(symvar_type="SX"):
    mdl = Model("discrete", symvar_type)
    x = mdl.set_variable(var_type="_x", var_name="x", shape=(4, 1))
    u = mdl.set_variable(var_type="_u", var_name="u", shape=(1, 1))
    mdl.set_expression(expr_name="cost", expr=sum1(x ** 2))
    x_next = mdl.set_variable(var_type="_z", var_name="x_next", shape=(4, 1))
    mdl.set_rhs("x", x_next)
    mdl.set_alg("x_next", x_next - A_D @ x - B_D @ u)
    mdl.setup()
    return mdl


def build_mpc(model, silence_solver=True, n_horizon=7, **overrides):
    mpc = MPC(model)
    st = mpc.settings
    st.n_robust, st.n_horizon, st.t_step = 0, n_horizon, 0.5
    st.store_full_solution = True
    for k, v in overrides.items():
        setattr(st, k, v)
    if silence_solver:
        st.supress_ipopt_output()
    mpc.set_objective(mterm=model.aux["cost"], lterm=model.aux["cost"])
    mpc.set_rterm(u=1e-4)
    limit = np.array([[4.0], [10.0], [4.0], [10.0]])
    mpc.bounds["lower", "_x", "x"] = -limit
    mpc.bounds["upper", "_x", "x"] = limit
    mpc.bounds["lower", "_u", "u"] = -0.5
    mpc.bounds["upper", "_u", "u"] = 0.5
    mpc.setup()
    return mpc
