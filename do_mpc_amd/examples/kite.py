"""Tethered kite, economic NMPC (maximise the tether force) with a soft minimum-height constraint.

Equations / tuning: /root/reference/examples/kite/template_model.py:34-98, template_mpc.py:34-103.  main.py:44-72 draws
the wind speed `w_ref`, the glide ratio `E_0`, the height limit `h_min` and the initial state at random; the values here
are fixed (the same ones as oracle/models.py:case_kite) so that the parity tests are reproducible.
"""
import numpy as np

from .. import MPC, Model
from ..sym import DM, cos, sin, sqrt, tan

W_REF, E_0, H_MIN = 10.0, 6.0, 100.0
L_TETHER, AREA, RHO, BETA, C_TILDE = 400.0, 300.0, 1.0, 0.0, 0.028


def build_model(symvar_type="SX"):
    mdl = Model("continuous", symvar_type)
    theta = mdl.set_variable("_x", "theta")            # zenith angle
    phi = mdl.set_variable("_x", "phi")                # azimuth angle
    psi = mdl.set_variable("_x", "psi")                # orientation of the kite
    u = mdl.set_variable("_u", "u_tilde")
    e0 = mdl.set_variable("_p", "E_0")
    v0 = mdl.set_variable("_p", "v_0")
    mdl.set_expression("E_0", e0)
    mdl.set_expression("v_0", v0)
    glide = e0 - C_TILDE * u ** 2
    v_a = v0 * glide * cos(theta)
    p_dyn = (RHO * v0 ** 2) / 2.0
    force = (p_dyn * AREA * cos(theta) ** 2 * (glide + 1.0) * sqrt(glide ** 2 + 1.0)) * (
        cos(theta) * np.cos(BETA) + sin(theta) * np.sin(BETA) * sin(phi))
    height = L_TETHER * sin(theta) * cos(phi)
    mdl.set_expression("T_F", force)
    mdl.set_expression("height_kite", height)
    dphi = -v_a / (L_TETHER * sin(theta)) * sin(psi)
    mdl.set_rhs("theta", v_a / L_TETHER * (cos(psi) - tan(theta) / glide))
    mdl.set_rhs("phi", dphi)
    mdl.set_rhs("psi", v_a / L_TETHER * u + dphi * (cos(theta)))
    mdl.setup()
    return mdl


def build_mpc(model, silence_solver=True, w_ref=W_REF, e_0=E_0, h_min=H_MIN, **overrides):
    mpc = MPC(model)
    st = mpc.settings
    st.n_horizon, st.n_robust, st.open_loop = 80, 0, 0
    st.t_step = 0.15
    st.store_full_solution = True
    for k, v in overrides.items():
        setattr(st, k, v)
    if silence_solver:
        st.supress_ipopt_output()
    mpc.set_objective(mterm=DM(0), lterm=-model.aux["T_F"] / 1e4)
    mpc.set_rterm(u_tilde=0.5)
    mpc.bounds["lower", "_x", "theta"] = 0.0
    mpc.bounds["lower", "_x", "phi"] = -0.5 * np.pi
    mpc.bounds["lower", "_x", "psi"] = -1.0 * np.pi
    mpc.bounds["upper", "_x", "theta"] = 0.5 * np.pi
    mpc.bounds["upper", "_x", "phi"] = 0.5 * np.pi
    mpc.bounds["upper", "_x", "psi"] = 1.0 * np.pi
    mpc.bounds["lower", "_u", "u_tilde"] = -10
    mpc.bounds["upper", "_u", "u_tilde"] = 10
    mpc.set_nl_cons("height_kite", -model.aux["height_kite"], ub=-h_min, soft_constraint=True,
                    penalty_term_cons=1e3, maximum_violation=10)
    mpc.set_uncertainty_values(E_0=np.array([e_0]), v_0=np.array([w_ref, w_ref * 0.8, w_ref * 1.2]))
    mpc.setup()
    return mpc


X0 = np.array([0.5, 0.3, 0.2])
