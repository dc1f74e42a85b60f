"""Industrial polymerisation reactor, robust multi-stage NMPC (BASELINE north-star workload).

Equations / tuning: /root/reference/examples/industrial_poly/template_model.py:32-134,
template_mpc.py:35-117, initial state main.py:59-73.
"""
import numpy as np

from .. import MPC, Model
from ..sym import exp

# physical constants
R_GAS, T_FEED, E_ACT, A_JACKET = 8.314, 298.15, 8500.0, 65.0
K_U1, K_U2 = 4.0, 32.0
W_WF, W_AF = 0.333, 0.667
M_M_KW, FM_M_KW = 5000.0, 300000.0
M_AWT_KW, FM_AWT_KW = 1000.0, 100000.0
M_AWT, FM_AWT, M_STEEL = 200.0, 20000.0, 39000.0
CP_W, CP_S, CP_F, CP_R = 4.2, 0.47, 3.0, 5.0
K_WS, K_AS, K_PS = 17280.0, 3600.0, 360.0
ALFA = 5 * 20e4 * 3.6
T_SET, T_BAND = 363.15, 2.0


def build_model(symvar_type="SX"):
    mdl = Model("continuous", symvar_type)
    s = {n: mdl.set_variable("_x", n) for n in
         ("m_W", "m_A", "m_P", "T_R", "T_S", "Tout_M", "T_EK", "Tout_AWT", "accum_monom", "T_adiab")}
    feed = mdl.set_variable("_u", "m_dot_f")
    T_jacket_in = mdl.set_variable("_u", "T_in_M")
    T_ehe_in = mdl.set_variable("_u", "T_in_EK")
    dH = mdl.set_variable("_p", "delH_R")
    k0 = mdl.set_variable("_p", "k_0")

    mW, mA, mP, TR, TS, TM, TEK, TAWT = (s[k] for k in ("m_W", "m_A", "m_P", "T_R", "T_S", "Tout_M", "T_EK", "Tout_AWT"))
    # (operations associated exactly like the reference template, so that the un-edited template_model.py lowers to the SAME
    #  generated header - and therefore the same gfx950 code object - as this restatement; tests/test_reference_templates.py)
    conv = mP / (mA + mP)
    m_tot = mW + mA + mP
    gel = (K_U1 * (1 - conv)) + (K_U2 * conv)
    k_reactor = k0 * exp(-E_ACT / (R_GAS * TR)) * gel
    k_loop = k0 * exp(-E_ACT / (R_GAS * TEK)) * ((K_U1 * (1 - conv)) + (K_U2 * conv))
    k_wall = ((mW / m_tot) * K_WS) + ((mA / m_tot) * K_AS) + ((mP / m_tot) * K_PS)
    P_1 = 1.0
    hold_up = mA - ((mA * M_AWT) / (mW + mA + mP))          # monomer outside the external heat exchanger

    d_mW = feed * W_WF
    mdl.set_rhs("m_W", d_mW)
    d_mA = (feed * W_AF) - (k_reactor * hold_up) - (P_1 * k_loop * (mA / m_tot) * M_AWT)
    mdl.set_rhs("m_A", d_mA)
    d_mP = (k_reactor * hold_up) + (P_1 * k_loop * (mA / m_tot) * M_AWT)
    mdl.set_rhs("m_P", d_mP)
    d_TR = 1. / (CP_R * m_tot) * ((feed * CP_F * (T_FEED - TR)) - (k_wall * A_JACKET * (TR - TS))
                                  - (FM_AWT * CP_R * (TR - TEK)) + (dH * k_reactor * hold_up))
    mdl.set_rhs("T_R", d_TR)
    mdl.set_rhs("T_S", 1. / (CP_S * M_STEEL) * ((k_wall * A_JACKET * (TR - TS)) - (k_wall * A_JACKET * (TS - TM))))
    mdl.set_rhs("Tout_M", 1. / (CP_W * M_M_KW) * ((FM_M_KW * CP_W * (T_jacket_in - TM)) + (k_wall * A_JACKET * (TS - TM))))
    mdl.set_rhs("T_EK", 1. / (CP_R * M_AWT) * ((FM_AWT * CP_R * (TR - TEK)) - (ALFA * (TEK - TAWT))
                                               + (P_1 * k_loop * (mA / m_tot) * M_AWT * dH)))
    mdl.set_rhs("Tout_AWT", 1. / (CP_W * M_AWT_KW) * ((FM_AWT_KW * CP_W * (T_ehe_in - TAWT)) - (ALFA * (TAWT - TEK))))
    mdl.set_rhs("accum_monom", feed)
    mdl.set_rhs("T_adiab", dH / (m_tot * CP_R) * d_mA - (d_mA + d_mW + d_mP) * (mA * dH / (m_tot * m_tot * CP_R)) + d_TR)
    mdl.setup()
    return mdl


def build_mpc(model, silence_solver=True, n_horizon=20, n_robust=1, uncertainty="product", **overrides):
    """uncertainty='product': the shipped 3x3 grid (9 combinations); 'paired': 3 combinations
    (nominal, +30 %, -30 % on both), used for the n_robust>=2 trees of BASELINE.json."""
    mpc = MPC(model)
    st = mpc.settings
    st.n_horizon, st.n_robust, st.open_loop = n_horizon, n_robust, 0
    st.t_step = 50.0 / 3600.0
    st.state_discretization = "collocation"
    st.store_full_solution = True
    for k, v in overrides.items():
        setattr(st, k, v)
    if silence_solver:
        st.supress_ipopt_output()
    mpc.set_objective(mterm=-model.x["m_P"], lterm=-model.x["m_P"])
    mpc.set_rterm(m_dot_f=0.002, T_in_M=0.004, T_in_EK=0.002)
    lower = dict(m_W=0.0, m_A=0.0, m_P=26.0, T_R=T_SET - T_BAND, T_S=298.0, Tout_M=298.0, T_EK=288.0,
                 Tout_AWT=288.0, accum_monom=0.0)
    upper = dict(T_R=T_SET + T_BAND, T_S=400.0, Tout_M=400.0, T_EK=400.0, Tout_AWT=400.0,
                 accum_monom=30000.0, T_adiab=382.15)
    for k, v in lower.items():
        mpc.bounds["lower", "_x", k] = v
    for k, v in upper.items():
        mpc.bounds["upper", "_x", k] = v
    for k, (lo, hi) in dict(m_dot_f=(0.0, 3.0e4), T_in_M=(333.15, 373.15), T_in_EK=(333.15, 373.15)).items():
        mpc.bounds["lower", "_u", k] = lo
        mpc.bounds["upper", "_u", k] = hi
    for k in ("m_W", "m_A", "m_P", "accum_monom"):
        mpc.scaling["_x", k] = 10
    mpc.scaling["_u", "m_dot_f"] = 100
    if st.n_robust == 0:
        mpc.set_nl_cons("T_R_UB", model.x["T_R"], ub=T_SET + T_BAND, soft_constraint=True, penalty_term_cons=1e4)
    else:
        mpc.bounds["upper", "_x", "T_R"] = T_SET + T_BAND
    dH_vals = np.array([950.0, 950.0 * 1.30, 950.0 * 0.70])
    k0_vals = np.array([7.0, 7.0 * 1.30, 7.0 * 0.70])
    if uncertainty == "product":
        mpc.set_uncertainty_values(delH_R=dH_vals, k_0=k0_vals)
    else:
        tmpl = mpc.get_p_template(3)
        for c in range(3):
            tmpl["_p", c, "delH_R"] = dH_vals[c]
            tmpl["_p", c, "k_0"] = k0_vals[c]
        mpc.set_p_fun(lambda t: tmpl)
    mpc.setup()
    return mpc


def _x0():
    x = np.array([10000.0, 853.0, 26.5, 363.15, 363.15, 363.15, 308.15, 308.15, 300.0, 0.0])
    x[9] = x[1] * 950.0 / ((x[0] + x[1] + x[2]) * 5.0) + x[3]
    return x


X0 = _x0()
