"""Kinematic and dynamic single-track (bicycle) vehicle models, nominal NMPC.

Equations / tuning: /root/reference/examples/kinematic_bicycle_model/template_model.py:34-75, template_mpc.py:34-95,
main.py:57-62 and /root/reference/examples/dynamic_bicycle_model/template_model.py:34-104, template_mpc.py:34-100,
main.py:57-64.  Two cases with trigonometric right-hand sides (tan / atan / atan2 / sin / cos): non-convex problems on
which the inertia correction of the factorisation is active.
"""
import numpy as np

from .. import MPC, Model
from ..sym import atan, atan2, cos, sin, sqrt, tan


class _Case:
    def __init__(self, build_model, build_mpc, x0):
        self.build_model, self.build_mpc, self.X0 = build_model, build_mpc, x0


def _settings(mpc, t_step, overrides, silence_solver):
    st = mpc.settings
    st.n_horizon, st.n_robust, st.open_loop = 10, 0, 0
    st.t_step = t_step
    st.state_discretization, st.collocation_type = "collocation", "radau"
    st.collocation_deg, st.collocation_ni = 2, 1
    st.store_full_solution = True
    for k, v in overrides.items():
        setattr(st, k, v)
    if silence_solver:
        st.supress_ipopt_output()


# ------------------------------------------------------------------------------------------------ kinematic
def kinematic_model(symvar_type="SX"):
    mdl = Model("continuous", symvar_type)
    lf, lr = 0.3, 0.3
    mdl.set_variable(var_type="_x", var_name="X_p", shape=(1, 1))
    mdl.set_variable(var_type="_x", var_name="Y_p", shape=(1, 1))
    psi = mdl.set_variable(var_type="_x", var_name="Psi", shape=(1, 1))
    v = mdl.set_variable(var_type="_x", var_name="V", shape=(1, 1))
    delta = mdl.set_variable(var_type="_u", var_name="Delta")
    acc = mdl.set_variable(var_type="_u", var_name="Acc")
    slip = atan((lr / (lr + lf)) * tan(delta))
    mdl.set_rhs("X_p", v * cos(psi + slip))
    mdl.set_rhs("Y_p", v * sin(psi + slip))
    mdl.set_rhs("Psi", (v / lr) * sin(slip))
    mdl.set_rhs("V", acc)
    mdl.setup()
    return mdl


def kinematic_mpc(model, silence_solver=True, **overrides):
    mpc = MPC(model)
    _settings(mpc, 0.05, overrides, silence_solver)
    x = model.x
    mpc.set_objective(mterm=(x["Y_p"] - 2) ** 2 + (x["X_p"] - 3) ** 2 + (x["Psi"] - 0) ** 2, lterm=(x["Y_p"] - 1) ** 2 * 0)
    mpc.set_rterm(Delta=1.0, Acc=1e-3)
    for name, b in dict(X_p=50.0, Y_p=50.0, Psi=np.pi / 2, V=5.0).items():
        mpc.bounds["lower", "_x", name] = -b
        mpc.bounds["upper", "_x", name] = b
    for name in ("Delta", "Acc"):
        mpc.bounds["lower", "_u", name] = -5
        mpc.bounds["upper", "_u", name] = 5
    mpc.setup()
    return mpc


# ------------------------------------------------------------------------------------------------ dynamic
def dynamic_model(symvar_type="SX"):
    mdl = Model("continuous", symvar_type)
    m, i_z, lf, lr = 5.692, 0.204, 0.178, 0.147
    d_f, d_r, c_f, c_r, b_f, b_r = 134.585, 159.919, 0.085, 0.133, 9.242, 17.716          # Pacejka tyre coefficients
    c_m1, c_m2, c_m3, c_m4 = 20, 6.92 * 1e-7, 3.99, 0.67                                    # drive-train map
    mdl.set_variable(var_type="_x", var_name="X_p", shape=(1, 1))
    mdl.set_variable(var_type="_x", var_name="Y_p", shape=(1, 1))
    psi = mdl.set_variable(var_type="_x", var_name="Psi", shape=(1, 1))
    vx = mdl.set_variable(var_type="_x", var_name="V_x", shape=(1, 1))
    vy = mdl.set_variable(var_type="_x", var_name="V_y", shape=(1, 1))
    w = mdl.set_variable(var_type="_x", var_name="W", shape=(1, 1))
    delta = mdl.set_variable(var_type="_u", var_name="Delta")
    pwm = mdl.set_variable(var_type="_u", var_name="d")
    mdl.set_expression(expr_name="Vel", expr=sqrt(vx ** 2 + vy ** 2))
    slip_f = -atan2(w * lf + vy, vx) + delta
    slip_r = atan2((w * lr - vy), vx)
    fy_f = d_f * sin(c_f * atan(b_f * slip_f))
    fy_r = d_r * sin(c_r * atan(b_r * slip_r))
    fx = (c_m1 - c_m2 * vx) * pwm - c_m4 * vx ** 2 - c_m3
    mdl.set_rhs("X_p", vx * cos(psi) - vy * sin(psi))
    mdl.set_rhs("Y_p", vx * sin(psi) + vy * cos(psi))
    mdl.set_rhs("Psi", w)
    mdl.set_rhs("V_x", (1 / m) * (fx - fy_f * sin(delta) + m * vy * w))
    mdl.set_rhs("V_y", (1 / m) * (fy_r + fy_f * cos(delta) - m * vx * w))
    mdl.set_rhs("W", (1 / i_z) * (fy_f * lf * cos(delta) - lf * fx * sin(delta) - lr * fy_r))
    mdl.setup()
    return mdl


def dynamic_mpc(model, silence_solver=True, **overrides):
    mpc = MPC(model)
    _settings(mpc, 0.1, overrides, silence_solver)
    cost = (model.x["Y_p"] - 1) ** 2
    mpc.set_objective(mterm=cost, lterm=cost)
    mpc.set_rterm(Delta=1e-3, d=1e-3)
    lo = dict(X_p=-50000, Y_p=-2, Psi=-0.78, V_x=0.1, V_y=-1, W=-0.2)
    hi = dict(X_p=50000, Y_p=2, Psi=0.78, V_x=5, V_y=1, W=0.2)
    for name in lo:
        mpc.bounds["lower", "_x", name] = lo[name]
        mpc.bounds["upper", "_x", name] = hi[name]
    mpc.bounds["lower", "_u", "Delta"] = -2
    mpc.bounds["upper", "_u", "Delta"] = 2
    mpc.bounds["lower", "_u", "d"] = 0
    mpc.bounds["upper", "_u", "d"] = 1
    mpc.setup()
    return mpc


kinematic = _Case(kinematic_model, kinematic_mpc, np.array([0.0, 0.0, 0.0, 0.1]))
dynamic = _Case(dynamic_model, dynamic_mpc, np.array([0.0, 0.0, 0.0, 0.1, 0.0, 0.0]))
