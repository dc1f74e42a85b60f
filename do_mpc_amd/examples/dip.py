"""Double inverted pendulum on a cart as a DAE: the accelerations are algebraic states defined by the Euler-Lagrange
equations; swing-up NMPC around an obstacle with uncertain rod masses declared (nominal scenario, n_robust = 0).

Equations / tuning: /root/reference/examples/double_inverted_pendulum/template_model.py:34-146, template_mpc.py:34-100;
obstacle and initial state of the reference's test: /root/reference/testing/test_DIP.py:70-90.
The reference's only continuous-time model with `_z` on the MPC path: collocation (Radau, degree 3) with algebraic rows at
every stored point, nl_cons (hard obstacle constraints), a `_tvp` set-point and nine declared (m1, m2) combinations.
"""
import numpy as np

from .. import MPC, Model
from ..sym import cos, sin, sqrt, vertcat

OBSTACLES = [{"x": 0.0, "y": 0.6, "r": 0.3}]
X0 = np.array([0.0, 0.9 * np.pi, 0.9 * np.pi, 0.0, 0.0, 0.0])


def build_model(obstacles=None, symvar_type="SX"):
    obstacles = OBSTACLES if obstacles is None else obstacles
    mdl = Model("continuous", symvar_type)
    m0, L1, L2 = 0.6, 0.5, 0.5
    l1, l2 = L1 / 2, L2 / 2
    m1_, m2_ = 0.2, 0.2
    J1 = (m1_ * l1 ** 2) / 3
    J2 = (m2_ * l2 ** 2) / 3
    m1 = mdl.set_variable("_p", "m1")
    m2 = mdl.set_variable("_p", "m2")
    g = 9.80665
    h1 = m0 + m1 + m2
    h2 = m1 * l1 + m2 * L1
    h3 = m2 * l2
    h4 = m1 * l1 ** 2 + m2 * L1 ** 2 + J1
    h5 = m2 * l2 * L1
    h6 = m2 * l2 ** 2 + J2
    h7 = (m1 * l1 + m2 * L1) * g
    h8 = m2 * l2 * g
    pos_set = mdl.set_variable("_tvp", "pos_set")
    pos = mdl.set_variable("_x", "pos")
    theta = mdl.set_variable("_x", "theta", (2, 1))
    dpos = mdl.set_variable("_x", "dpos")
    dtheta = mdl.set_variable("_x", "dtheta", (2, 1))
    ddpos = mdl.set_variable("_z", "ddpos")
    ddtheta = mdl.set_variable("_z", "ddtheta", (2, 1))
    u = mdl.set_variable("_u", "force")
    mdl.set_rhs("pos", dpos)
    mdl.set_rhs("theta", dtheta)
    mdl.set_rhs("dpos", ddpos)
    mdl.set_rhs("dtheta", ddtheta)
    euler_lagrange = vertcat(
        h1 * ddpos + h2 * ddtheta[0] * cos(theta[0]) + h3 * ddtheta[1] * cos(theta[1])
        - (h2 * dtheta[0] ** 2 * sin(theta[0]) + h3 * dtheta[1] ** 2 * sin(theta[1]) + u),
        h2 * cos(theta[0]) * ddpos + h4 * ddtheta[0] + h5 * cos(theta[0] - theta[1]) * ddtheta[1]
        - (h7 * sin(theta[0]) - h5 * dtheta[1] ** 2 * sin(theta[0] - theta[1])),
        h3 * cos(theta[1]) * ddpos + h5 * cos(theta[0] - theta[1]) * ddtheta[0] + h6 * ddtheta[1]
        - (h5 * dtheta[0] ** 2 * sin(theta[0] - theta[1]) + h8 * sin(theta[1])))
    mdl.set_alg("euler_lagrange", euler_lagrange)
    E_kin_cart = 1 / 2 * m0 * dpos ** 2
    E_kin_p1 = 1 / 2 * m1 * ((dpos + l1 * dtheta[0] * cos(theta[0])) ** 2 + (l1 * dtheta[0] * sin(theta[0])) ** 2) \
        + 1 / 2 * J1 * dtheta[0] ** 2
    E_kin_p2 = 1 / 2 * m2 * ((dpos + L1 * dtheta[0] * cos(theta[0]) + l2 * dtheta[1] * cos(theta[1])) ** 2
                             + (L1 * dtheta[0] * sin(theta[0]) + l2 * dtheta[1] * sin(theta[1])) ** 2) \
        + 1 / 2 * J2 * dtheta[0] ** 2
    E_kin = E_kin_cart + E_kin_p1 + E_kin_p2
    E_pot = m1 * g * l1 * cos(theta[0]) + m2 * g * (L1 * cos(theta[0]) + l2 * cos(theta[1]))
    mdl.set_expression("E_kin", E_kin)
    mdl.set_expression("E_pot", E_pot)
    node0_x, node0_y = mdl.x["pos"], 0.0
    node1_x = node0_x + L1 * sin(mdl.x["theta", 0])
    node1_y = node0_y + L1 * cos(mdl.x["theta", 0])
    node2_x = node1_x + L2 * sin(mdl.x["theta", 1])
    node2_y = node1_y + L2 * cos(mdl.x["theta", 1])
    dist = []
    for obs in obstacles:
        for nx_, ny_ in ((node0_x, node0_y), (node1_x, node1_y), (node2_x, node2_y)):
            dist.append(sqrt((nx_ - obs["x"]) ** 2 + (ny_ - obs["y"]) ** 2) - obs["r"] * 1.05)
    mdl.set_expression("obstacle_distance", vertcat(*dist))
    mdl.set_expression("tvp", pos_set)
    mdl.setup()
    return mdl


def build_mpc(model, silence_solver=True, **overrides):
    mpc = MPC(model)
    st = mpc.settings
    st.n_horizon, st.n_robust, st.open_loop, st.t_step = 100, 0, 0, 0.04
    st.state_discretization, st.collocation_type, st.collocation_deg, st.collocation_ni = "collocation", "radau", 3, 1
    st.store_full_solution = True
    for k, v in overrides.items():
        setattr(st, k, v)
    if silence_solver:
        st.supress_ipopt_output()
    mterm = model.aux["E_kin"] - model.aux["E_pot"]
    lterm = -model.aux["E_pot"] + 10 * (model.x["pos"] - model.tvp["pos_set"]) ** 2
    mpc.set_objective(mterm=mterm, lterm=lterm)
    mpc.set_rterm(force=0.1)
    mpc.bounds["lower", "_u", "force"] = -4
    mpc.bounds["upper", "_u", "force"] = 4
    mpc.set_nl_cons("obstacles", -model.aux["obstacle_distance"], 0)
    m_var = 0.2 * np.array([1, 0.95, 1.05])
    mpc.set_uncertainty_values(m1=m_var, m2=m_var)
    tvp_template = mpc.get_tvp_template()
    ind_switch = 4 // st.t_step

    def tvp_fun(t_ind):
        ind = t_ind // st.t_step
        tvp_template["_tvp", :, "pos_set"] = -0.8 if ind <= ind_switch else 0.8
        return tvp_template

    mpc.set_tvp_fun(tvp_fun)
    mpc.setup()
    return mpc
