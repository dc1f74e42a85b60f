"""Continuous stirred tank reactor NMPC.

Equations / tuning: /root/reference/examples/CSTR/template_model.py:34-98,
template_mpc.py:34-105, initial state main.py:58-63.
"""
import numpy as np

from .. import MPC, Model
from ..sym import exp

K0_AB = K0_BC = 1.287e12
K0_AD = 9.043e9
EA_AB = EA_BC = 9758.3
EA_AD = 8560.0
H_AB, H_BC, H_AD = 4.2, -11.0, -41.85
RHO, CP, CP_K, A_R, V_R, M_K, T_IN, K_W = 0.9342, 3.01, 2.0, 0.215, 10.01, 5.0, 130.0, 4032.0
C_A0 = (5.7 + 4.5) / 2.0


def build_model(symvar_type="SX"):
    mdl = Model("continuous", symvar_type)
    Ca = mdl.set_variable(var_type="_x", var_name="C_a", shape=(1, 1))
    Cb = mdl.set_variable(var_type="_x", var_name="C_b", shape=(1, 1))
    Tr = mdl.set_variable(var_type="_x", var_name="T_R", shape=(1, 1))
    Tk = mdl.set_variable(var_type="_x", var_name="T_K", shape=(1, 1))
    F = mdl.set_variable(var_type="_u", var_name="F")
    Qdot = mdl.set_variable(var_type="_u", var_name="Q_dot")
    alpha = mdl.set_variable(var_type="_p", var_name="alpha")
    beta = mdl.set_variable(var_type="_p", var_name="beta")
    dT = mdl.set_expression(expr_name="T_dif", expr=Tr - Tk)
    k1 = beta * K0_AB * exp(-EA_AB / (Tr + 273.15))
    k2 = K0_BC * exp(-EA_BC / (Tr + 273.15))
    k3 = K0_AD * exp(-alpha * EA_AD / (Tr + 273.15))
    mdl.set_rhs("C_a", F * (C_A0 - Ca) - k1 * Ca - k3 * (Ca ** 2))
    mdl.set_rhs("C_b", -F * Cb + k1 * Ca - k2 * Cb)
    mdl.set_rhs("T_R", ((k1 * Ca * H_AB + k2 * Cb * H_BC + k3 * (Ca ** 2) * H_AD) / (-RHO * CP)) + F * (T_IN - Tr)
                + (((K_W * A_R) * (-dT)) / (RHO * CP * V_R)))
    mdl.set_rhs("T_K", (Qdot + K_W * A_R * dT) / (M_K * CP_K))
    mdl.setup()
    return mdl


def build_mpc(model, silence_solver=True, n_horizon=20, n_robust=1, collocation_deg=2, track_sign=1.0, custom_rterm=None, uncertainty=None, soft_T_R=True, **overrides):
    """track_sign=-1 turns the tracking cost into a concave one (C_b is pushed AWAY from 0.6, bounded only by the
    box): a non-convex test problem whose reduced Hessian needs inertia correction in most iterations."""
    mpc = MPC(model)
    st = mpc.settings
    st.n_horizon, st.n_robust, st.open_loop = n_horizon, n_robust, 0
    st.t_step = 0.005
    st.state_discretization, st.collocation_type = "collocation", "radau"
    st.collocation_deg, st.collocation_ni = collocation_deg, 1
    st.store_full_solution = True
    for k, v in overrides.items():
        setattr(st, k, v)
    if silence_solver:
        st.supress_ipopt_output()
    mpc.scaling["_x", "T_R"] = 100
    mpc.scaling["_x", "T_K"] = 100
    mpc.scaling["_u", "Q_dot"] = 2000
    mpc.scaling["_u", "F"] = 100
    track = track_sign * (model.x["C_b"] - 0.6) ** 2
    mpc.set_objective(mterm=track, lterm=track)
    if custom_rterm is not None:          # user-defined input penalty: expression in model.x / model.u / mpc.u_prev (_mpc.py:593-677)
        mpc.set_rterm(rterm=(RTERM_VARIANTS[custom_rterm] if isinstance(custom_rterm, str) else custom_rterm)(model, mpc))
    else:
        mpc.set_rterm(F=0.1, Q_dot=1e-3)
    for k, v in dict(C_a=0.1, C_b=0.1, T_R=50, T_K=50).items():
        mpc.bounds["lower", "_x", k] = v
    for k, v in dict(C_a=2, C_b=2, T_K=140).items():
        mpc.bounds["upper", "_x", k] = v
    mpc.bounds["lower", "_u", "F"] = 5
    mpc.bounds["lower", "_u", "Q_dot"] = -8500
    mpc.bounds["upper", "_u", "F"] = 100
    mpc.bounds["upper", "_u", "Q_dot"] = 0.0
    mpc.set_nl_cons("T_R", model.x["T_R"], ub=140, soft_constraint=bool(soft_T_R), penalty_term_cons=1e2)
    if uncertainty is None:
        uncertainty = dict(alpha=[1.0, 1.05, 0.95], beta=[1.0, 1.1, 0.9])
    mpc.set_uncertainty_values(**{k: np.asarray(v, float) for k, v in uncertainty.items()})
    mpc.setup()
    return mpc


X0 = np.array([0.8, 0.5, 134.14, 130.0])


def _rterm_default_as_expression(model, mpc):
    """the default quadratic penalty written as an expression: r' (u / u_scaling - u_prev)^2 - unscaled u, SCALED u_prev
    (the reference evaluates a user-defined rterm that way, _mpc.py:1263-1269)"""
    return 0.1 * (model.u["F"] / 100.0 - mpc.u_prev["F"]) ** 2 + 1e-3 * (model.u["Q_dot"] / 2000.0 - mpc.u_prev["Q_dot"]) ** 2


def _rterm_custom(model, mpc):
    """not a quadratic, depends on a state: weights the input move by the reactor temperature, quartic term, input coupling"""
    dF = model.u["F"] / 100.0 - mpc.u_prev["F"]
    dQ = model.u["Q_dot"] / 2000.0 - mpc.u_prev["Q_dot"]
    return 0.1 * (1 + 0.01 * model.x["T_R"]) * dF ** 2 + 1e-3 * dQ ** 2 + 0.5 * dF ** 4 + 0.02 * dF * dQ


RTERM_VARIANTS = {"default_as_expression": _rterm_default_as_expression, "custom": _rterm_custom}
