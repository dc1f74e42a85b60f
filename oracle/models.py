"""SymPy restatement of the four BASELINE example problems (TEST INFRASTRUCTURE).

Independent of the product's own expression DAG (do_mpc_amd/sym.py): derivatives
here come from sympy.diff, so AD bugs in the product cannot hide in the oracle.

Each `case_*` returns a dict describing exactly what the reference templates
configure:
  industrial_poly : /root/reference/examples/industrial_poly/template_model.py:32-134,
                    template_mpc.py:35-117, main.py:59-73
  CSTR            : /root/reference/examples/CSTR/template_model.py:34-98,
                    template_mpc.py:34-105, main.py:58-63
  batch_reactor   : /root/reference/examples/batch_reactor/template_model.py:34-74,
                    template_mpc.py:34-88, main.py:56-61
  oscillating_masses_discrete : /root/reference/examples/oscillating_masses_discrete/
                    template_model.py:34-73, template_mpc.py:34-74, main.py:57-60
"""
import itertools

import numpy as np
import sympy as sp

INF = np.inf


def _base(**kw):
    d = dict(model_type="continuous", n_robust=0, open_loop=False, collocation_type="radau",
             collocation_deg=2, collocation_ni=1, cons_check_colloc_points=True,
             nl_cons_check_colloc_points=False, nl_cons_single_slack=False,
             use_terminal_bounds=False, nl_cons=[], uncertainty={}, tvp_names=[])
    d.update(kw)
    return d


def case_industrial_poly(**over):
    x = sp.symbols("m_W m_A m_P T_R T_S Tout_M T_EK Tout_AWT accum_monom T_adiab")
    u = sp.symbols("m_dot_f T_in_M T_in_EK")
    p = sp.symbols("delH_R k_0")
    m_W, m_A, m_P, T_R, T_S, Tout_M, T_EK, Tout_AWT, accum_monom, T_adiab = x
    m_dot_f, T_in_M, T_in_EK = u
    delH_R, k_0 = p
    R, T_F, E_a, A_tank = 8.314, 25 + 273.15, 8500.0, 65.0
    k_U2, k_U1, w_WF, w_AF = 32.0, 4.0, .333, .667
    m_M_KW, fm_M_KW, m_AWT_KW, fm_AWT_KW = 5000.0, 300000.0, 1000.0, 100000.0
    m_AWT, fm_AWT, m_S = 200.0, 20000.0, 39000.0
    c_pW, c_pS, c_pF, c_pR = 4.2, .47, 3.0, 5.0
    k_WS, k_AS, k_PS = 17280.0, 3600.0, 360.0
    alfa = 5 * 20e4 * 3.6
    p_1 = 1.0
    U_m = m_P / (m_A + m_P)
    m_ges = m_W + m_A + m_P
    k_R1 = k_0 * sp.exp(-E_a / (R * T_R)) * ((k_U1 * (1 - U_m)) + (k_U2 * U_m))
    k_R2 = k_0 * sp.exp(-E_a / (R * T_EK)) * ((k_U1 * (1 - U_m)) + (k_U2 * U_m))
    k_K = ((m_W / m_ges) * k_WS) + ((m_A / m_ges) * k_AS) + ((m_P / m_ges) * k_PS)
    dot_m_W = m_dot_f * w_WF
    reac = k_R1 * (m_A - ((m_A * m_AWT) / (m_W + m_A + m_P)))
    ehe = p_1 * k_R2 * (m_A / m_ges) * m_AWT
    dot_m_A = (m_dot_f * w_AF) - reac - ehe
    dot_m_P = reac + ehe
    dot_T_R = 1. / (c_pR * m_ges) * ((m_dot_f * c_pF * (T_F - T_R)) - (k_K * A_tank * (T_R - T_S))
                                     - (fm_AWT * c_pR * (T_R - T_EK)) + (delH_R * reac))
    rhs = [
        dot_m_W, dot_m_A, dot_m_P, dot_T_R,
        1. / (c_pS * m_S) * ((k_K * A_tank * (T_R - T_S)) - (k_K * A_tank * (T_S - Tout_M))),
        1. / (c_pW * m_M_KW) * ((fm_M_KW * c_pW * (T_in_M - Tout_M)) + (k_K * A_tank * (T_S - Tout_M))),
        1. / (c_pR * m_AWT) * ((fm_AWT * c_pR * (T_R - T_EK)) - (alfa * (T_EK - Tout_AWT)) + (ehe * delH_R)),
        1. / (c_pW * m_AWT_KW) * ((fm_AWT_KW * c_pW * (T_in_EK - Tout_AWT)) - (alfa * (Tout_AWT - T_EK))),
        m_dot_f,
        delH_R / (m_ges * c_pR) * dot_m_A - (dot_m_A + dot_m_W + dot_m_P) * (m_A * delH_R / (m_ges * m_ges * c_pR)) + dot_T_R,
    ]
    tr = 2.0
    x_lb = np.array([0.0, 0.0, 26.0, 363.15 - tr, 298.0, 298.0, 288.0, 288.0, 0.0, -INF])
    x_ub = np.array([INF, INF, INF, 363.15 + tr, 400.0, 400.0, 400.0, 400.0, 30000.0, 382.15])
    x0 = np.array([10000.0, 853.0, 26.5, 363.15, 363.15, 363.15, 308.15, 308.15, 300.0, 0.0])
    x0[9] = x0[1] * 950.0 / ((x0[0] + x0[1] + x0[2]) * 5.0) + x0[3]
    d = _base(name="industrial_poly", x=x, u=u, p=p, rhs=rhs, lterm=-m_P, mterm=-m_P,
              rterm=np.array([0.002, 0.004, 0.002]), n_horizon=20, n_robust=1, t_step=50.0 / 3600.0,
              x_lb=x_lb, x_ub=x_ub, u_lb=np.array([0.0, 333.15, 333.15]), u_ub=np.array([3.0e4, 373.15, 373.15]),
              x_scaling=np.array([10., 10, 10, 1, 1, 1, 1, 1, 10, 1]), u_scaling=np.array([100., 1, 1]),
              uncertainty=dict(delH_R=[950.0, 950.0 * 1.30, 950.0 * 0.70], k_0=[7.0 * 1.00, 7.0 * 1.30, 7.0 * 0.70]),
              x0=x0, aux={})
    d.update(over)
    return d


def case_CSTR(**over):
    x = sp.symbols("C_a C_b T_R T_K")
    u = sp.symbols("F Q_dot")
    p = sp.symbols("alpha beta")
    C_a, C_b, T_R, T_K = x
    F, Q_dot = u
    alpha, beta = p
    K0_ab, K0_bc, K0_ad = 1.287e12, 1.287e12, 9.043e9
    E_A_ab, E_A_bc, E_A_ad = 9758.3, 9758.3, 8560.0
    H_R_ab, H_R_bc, H_R_ad = 4.2, -11.0, -41.85
    Rou, Cp, Cp_k, A_R, V_R, m_k, T_in, K_w = 0.9342, 3.01, 2.0, 0.215, 10.01, 5.0, 130.0, 4032.0
    C_A0 = (5.7 + 4.5) / 2.0
    T_dif = T_R - T_K
    K_1 = beta * K0_ab * sp.exp((-E_A_ab) / (T_R + 273.15))
    K_2 = K0_bc * sp.exp((-E_A_bc) / (T_R + 273.15))
    K_3 = K0_ad * sp.exp((-alpha * E_A_ad) / (T_R + 273.15))
    rhs = [
        F * (C_A0 - C_a) - K_1 * C_a - K_3 * (C_a ** 2),
        -F * C_b + K_1 * C_a - K_2 * C_b,
        ((K_1 * C_a * H_R_ab + K_2 * C_b * H_R_bc + K_3 * (C_a ** 2) * H_R_ad) / (-Rou * Cp)) + F * (T_in - T_R)
        + (((K_w * A_R) * (-T_dif)) / (Rou * Cp * V_R)),
        (Q_dot + K_w * A_R * T_dif) / (m_k * Cp_k),
    ]
    d = _base(name="CSTR", x=x, u=u, p=p, rhs=rhs, lterm=(C_b - 0.6) ** 2, mterm=(C_b - 0.6) ** 2,
              rterm=np.array([0.1, 1e-3]), n_horizon=20, n_robust=1, t_step=0.005,
              x_lb=np.array([0.1, 0.1, 50.0, 50.0]), x_ub=np.array([2.0, 2.0, INF, 140.0]),
              u_lb=np.array([5.0, -8500.0]), u_ub=np.array([100.0, 0.0]),
              x_scaling=np.array([1.0, 1.0, 100.0, 100.0]), u_scaling=np.array([100.0, 2000.0]),
              nl_cons=[dict(name="T_R", expr=T_R, ub=140.0, soft=True, penalty=1e2, max_violation=INF)],
              uncertainty=dict(alpha=[1., 1.05, 0.95], beta=[1., 1.1, 0.9]),
              x0=np.array([0.8, 0.5, 134.14, 130.0]), aux={"T_dif": T_dif})
    d.update(over)
    return d


def case_batch_reactor(**over):
    x = sp.symbols("X_s S_s P_s V_s")
    u = (sp.Symbol("inp"),)
    p = sp.symbols("Y_x S_in")
    X_s, S_s, P_s, V_s = x
    inp = u[0]
    Y_x, S_in = p
    mu_m, K_m, K_i, v_par, Y_p = 0.02, 0.05, 5.0, 0.004, 1.2
    mu_S = mu_m * S_s / (K_m + S_s + (S_s ** 2 / K_i))
    rhs = [mu_S * X_s - inp / V_s * X_s,
           -mu_S * X_s / Y_x - v_par * X_s / Y_p + inp / V_s * (S_in - S_s),
           v_par * X_s - inp / V_s * P_s,
           inp]
    d = _base(name="batch_reactor", x=x, u=u, p=p, rhs=rhs, lterm=-P_s, mterm=-P_s,
              rterm=np.array([1.0]), n_horizon=20, n_robust=0, t_step=1.0, collocation_ni=2,
              x_lb=np.array([0.0, -0.01, 0.0, 0.0]), x_ub=np.array([3.7, INF, 3.0, INF]),
              u_lb=np.array([0.0]), u_ub=np.array([0.2]),
              x_scaling=np.ones(4), u_scaling=np.ones(1),
              uncertainty=dict(Y_x=[0.5, 0.4, 0.3], S_in=[200.0, 220.0, 180.0]),
              x0=np.array([1.0, 0.5, 0.0, 120.0]), aux={})
    d.update(over)
    return d


def case_oscillating_masses(**over):
    x = sp.symbols("x_0 x_1 x_2 x_3")
    u = (sp.Symbol("u"),)
    A = np.array([[0.763, 0.460, 0.115, 0.020],
                  [-0.899, 0.763, 0.420, 0.115],
                  [0.115, 0.020, 0.763, 0.460],
                  [0.420, 0.115, -0.899, 0.763]])
    B = np.array([0.014, 0.063, 0.221, 0.367])
    rhs = [sum(A[i, j] * x[j] for j in range(4)) + B[i] * u[0] for i in range(4)]
    cost = sum(xi ** 2 for xi in x)
    mx = np.array([4.0, 10.0, 4.0, 10.0])
    rng = np.random.RandomState(99)  # reference test: np.random.seed(99); np.random.rand(4)-0.5
    x0 = rng.rand(4) - 0.5
    d = _base(name="oscillating_masses", model_type="discrete", x=x, u=u, p=(), rhs=rhs,
              lterm=cost, mterm=cost, rterm=np.array([1e-4]), n_horizon=7, n_robust=0, t_step=0.5,
              x_lb=-mx, x_ub=mx, u_lb=np.array([-0.5]), u_ub=np.array([0.5]),
              x_scaling=np.ones(4), u_scaling=np.ones(1), x0=x0, aux={"cost": cost})
    d.update(over)
    return d


def case_kinematic_bicycle(**over):
    """/root/reference/examples/kinematic_bicycle_model/template_model.py:34-75, template_mpc.py:34-95, main.py:57-62."""
    x = sp.symbols("X_p Y_p Psi V")
    u = sp.symbols("Delta Acc")
    X_p, Y_p, Psi, V = x
    Delta, Acc = u
    lf, lr = 0.3, 0.3
    Beta = sp.atan((lr / (lr + lf)) * sp.tan(Delta))
    rhs = [V * sp.cos(Psi + Beta), V * sp.sin(Psi + Beta), (V / lr) * sp.sin(Beta), Acc]
    mterm = (Y_p - 2) ** 2 + (X_p - 3) ** 2 + (Psi - 0) ** 2
    d = _base(name="kinematic_bicycle", x=x, u=u, p=(), rhs=rhs, lterm=sp.Integer(0), mterm=mterm,
              rterm=np.array([1.0, 1e-3]), n_horizon=10, n_robust=0, t_step=0.05,
              x_lb=np.array([-50.0, -50.0, -np.pi / 2, -5.0]), x_ub=np.array([50.0, 50.0, np.pi / 2, 5.0]),
              u_lb=np.array([-5.0, -5.0]), u_ub=np.array([5.0, 5.0]),
              x_scaling=np.ones(4), u_scaling=np.ones(2), x0=np.array([0.0, 0.0, 0.0, 0.1]), aux={})
    d.update(over)
    return d


def case_dynamic_bicycle(**over):
    """/root/reference/examples/dynamic_bicycle_model/template_model.py:34-104, template_mpc.py:34-100, main.py:57-64."""
    x = sp.symbols("X_p Y_p Psi V_x V_y W")
    u = sp.symbols("Delta d")
    X_p, Y_p, Psi, V_x, V_y, W = x
    Delta, dd = u
    m, I_z, lf, lr = 5.692, 0.204, 0.178, 0.147
    D_f, D_r, C_f, C_r, B_f, B_r = 134.585, 159.919, 0.085, 0.133, 9.242, 17.716
    c_m1, c_m2, c_m3, c_m4 = 20, 6.92 * 1e-7, 3.99, 0.67
    alpha_f = -sp.atan2(W * lf + V_y, V_x) + Delta
    alpha_r = sp.atan2((W * lr - V_y), V_x)
    F_f_y = D_f * sp.sin(C_f * sp.atan(B_f * alpha_f))
    F_r_y = D_r * sp.sin(C_r * sp.atan(B_r * alpha_r))
    F_x = (c_m1 - c_m2 * V_x) * dd - c_m4 * V_x ** 2 - c_m3
    rhs = [V_x * sp.cos(Psi) - V_y * sp.sin(Psi),
           V_x * sp.sin(Psi) + V_y * sp.cos(Psi),
           W,
           (1 / m) * (F_x - F_f_y * sp.sin(Delta) + m * V_y * W),
           (1 / m) * (F_r_y + F_f_y * sp.cos(Delta) - m * V_x * W),
           (1 / I_z) * (F_f_y * lf * sp.cos(Delta) - lf * F_x * sp.sin(Delta) - lr * F_r_y)]
    cost = (Y_p - 1) ** 2
    d = _base(name="dynamic_bicycle", x=x, u=u, p=(), rhs=rhs, lterm=cost, mterm=cost,
              rterm=np.array([1e-3, 1e-3]), n_horizon=10, n_robust=0, t_step=0.1,
              x_lb=np.array([-50000.0, -2.0, -0.78, 0.1, -1.0, -0.2]), x_ub=np.array([50000.0, 2.0, 0.78, 5.0, 1.0, 0.2]),
              u_lb=np.array([-2.0, 0.0]), u_ub=np.array([2.0, 1.0]),
              x_scaling=np.ones(6), u_scaling=np.ones(2), x0=np.array([0.0, 0.0, 0.0, 0.1, 0.0, 0.0]),
              aux={"Vel": sp.sqrt(V_x ** 2 + V_y ** 2)})
    d.update(over)
    return d


KITE = dict(w_ref=10.0, E_0=6.0, h_min=100.0)       # main.py:46-48 draws these at random; fixed here


def case_kite(**over):
    """/root/reference/examples/kite/template_model.py:34-98, template_mpc.py:34-103, main.py:44-72 (tethered kite, economic
    NMPC with a soft height constraint; w_ref, E_0, h_min and x0 are random draws in main.py, fixed values here)."""
    x = sp.symbols("theta phi psi")
    u = (sp.Symbol("u_tilde"),)
    p = sp.symbols("E_0 v_0")
    theta, phi, psi = x
    u_tilde = u[0]
    E_0, v_0 = p
    L_tether, A, rho, beta, c_tilde = 400.0, 300.0, 1.0, 0.0, 0.028
    E = E_0 - c_tilde * u_tilde ** 2
    v_a = v_0 * E * sp.cos(theta)
    P_D = (rho * v_0 ** 2) / 2.0
    T_F = (P_D * A * sp.cos(theta) ** 2 * (E + 1.0) * sp.sqrt(E ** 2 + 1.0)) * (
        sp.cos(theta) * np.cos(beta) + sp.sin(theta) * np.sin(beta) * sp.sin(phi))
    height = L_tether * sp.sin(theta) * sp.cos(phi)
    dphi = -v_a / (L_tether * sp.sin(theta)) * sp.sin(psi)
    rhs = [v_a / L_tether * (sp.cos(psi) - sp.tan(theta) / E), dphi, v_a / L_tether * u_tilde + dphi * sp.cos(theta)]
    w = KITE["w_ref"]
    d = _base(name="kite", x=x, u=u, p=p, rhs=rhs, lterm=-T_F / 1e4, mterm=sp.Integer(0),
              rterm=np.array([0.5]), n_horizon=80, n_robust=0, t_step=0.15,
              x_lb=np.array([0.0, -0.5 * np.pi, -np.pi]), x_ub=np.array([0.5 * np.pi, 0.5 * np.pi, np.pi]),
              u_lb=np.array([-10.0]), u_ub=np.array([10.0]),
              x_scaling=np.ones(3), u_scaling=np.ones(1),
              nl_cons=[dict(name="height_kite", expr=-height, ub=-KITE["h_min"], soft=True, penalty=1e3, max_violation=10.0)],
              uncertainty=dict(E_0=[KITE["E_0"]], v_0=[w, w * 0.8, w * 1.2]),
              x0=np.array([0.5, 0.3, 0.2]), aux={"E_0": E_0, "v_0": v_0, "T_F": T_F, "height_kite": height})
    d.update(over)
    return d


def case_rotating_masses(**over):
    """/root/reference/examples/rotating_oscillating_masses_mhe_mpc/template_model.py:34-99, template_mpc.py:34-106:
    three spring-coupled discs, two motors, set-point for the middle disc as a time-varying parameter.  The 5x5 `_tvp`
    P_v and the `_p` P_p belong to the example's estimator; declared so that the parameter vector has the same layout."""
    x = sp.symbols("phi_1 phi_2 phi_3 dphi_0 dphi_1 dphi_2 phi_m_0 phi_m_1")
    u = sp.symbols("phi_m_set_0 phi_m_set_1")
    p = sp.symbols("P_p Theta_1 Theta_2 Theta_3")
    tvp = (sp.Symbol("phi_2_set"),) + sp.symbols("P_v_0:25")
    phi, dphi, phi_m = x[0:3], x[3:6], x[6:8]
    th = p[1:]
    c = np.array([2.697, 2.66, 3.05, 2.86]) * 1e-3
    d = np.array([6.78, 8.01, 8.82]) * 1e-5
    left, right = [phi_m[0], phi[0], phi[1]], [phi[1], phi[2], phi_m[1]]
    rhs = list(dphi) + [-c[i] / th[i] * (phi[i] - left[i]) - c[i + 1] / th[i] * (phi[i] - right[i]) - d[i] / th[i] * dphi[i]
                        for i in range(3)] + [1 / 1e-2 * (u[i] - phi_m[i]) for i in range(2)]
    dd = _base(name="rotating_masses", x=x, u=u, p=p, tvp=tvp, rhs=rhs, lterm=(phi[1] - tvp[0]) ** 2, mterm=sp.Integer(1),
               rterm=np.array([1e-2, 1e-2]), n_horizon=20, n_robust=0, t_step=0.1,
               x_lb=-np.inf * np.ones(8), x_ub=np.inf * np.ones(8), u_lb=-5.0 * np.ones(2), u_ub=5.0 * np.ones(2),
               x_scaling=np.ones(8), u_scaling=np.ones(2), x0=np.zeros(8), aux={},
               uncertainty={"Theta_1": 2.25e-4 * np.array([1.0, 1.1]), "Theta_2": 2.25e-4 * np.array([1.0]),
                            "Theta_3": 2.25e-4 * np.array([1.0])})
    dd.update(over)
    return dd


def case_rotating_masses_mhe(**over):
    """The estimator of the same example: /root/reference/examples/rotating_oscillating_masses_mhe_mpc/template_mhe.py:34-104
    (MHE with `Theta_1` estimated, default objective with the 5x5 `_tvp` P_v and the `_p` P_p as weights, bounds on the inputs
    and the angular velocities, the box of Theta_1 as two nl_cons rows checked at the collocation points) on the model of
    template_model.py:44-99 (measurements: the three disc angles and the two motor set-points, each with measurement noise)."""
    d = case_rotating_masses()
    x, u, p, tvp = d["x"], d["u"], d["p"], d["tvp"]
    v = sp.symbols("v_0:5")
    xp = sp.symbols("xprev_0:8")
    pp = (sp.Symbol("Theta_1_prev"),)
    meas = [x[0] + v[0], x[1] + v[1], x[2] + v[2], u[0] + v[3], u[1] + v[4]]
    Pv = sp.Matrix(5, 5, lambda i, j: tvp[1 + i + 5 * j])             # (5, 5) entry of the `_tvp` struct, stored column by column
    vv = sp.Matrix(v)
    stage = (vv.T * Pv * vv)[0, 0]
    dx = sp.Matrix([x[i] - xp[i] for i in range(8)])
    arrival = 1e-4 * (dx.T * dx)[0, 0] + p[0] * (p[1] - pp[0]) ** 2    # P_x = 1e-4 I, P_p = the parameter `P_p`
    x_lb, x_ub = -np.inf * np.ones(8), np.inf * np.ones(8)
    x_lb[3:6], x_ub[3:6] = -6.0, 6.0
    dd = dict(d)
    dd.update(name="rotating_masses_mhe", v=v, w=(), meas=meas, p_est=[p[1]], x_prev=xp, p_est_prev=pp,
              stage_cost=stage, arrival_cost=arrival, n_horizon=10, t_step=0.1, collocation_deg=2, collocation_ni=1,
              nl_cons_check_colloc_points=True,
              nl_cons=[dict(name="p_est_lb", expr=-p[1] + 1e-5, ub=0.0, soft=False), dict(name="p_est_ub", expr=p[1] - 1e-3, ub=0.0, soft=False)],
              x_lb=x_lb, x_ub=x_ub, u_lb=-5.0 * np.ones(2), u_ub=5.0 * np.ones(2))
    dd.update(over)
    return dd


def case_rotating_masses_mhe_w(**over):
    """A second estimator on the same model, for the paths the shipped example leaves out (no stored run exists for it): process
    noise on the three angular velocities (`set_rhs(..., process_noise=True)`), weights as numbers, the box of Theta_1 as bounds of
    `_p_est`, one nl_cons row on a state checked at the states only (nl_cons_check_colloc_points = False), horizon 6."""
    d = case_rotating_masses_mhe()
    x, p = d["x"], d["p"]
    w = sp.symbols("w_0:3")
    rhs = list(d["rhs"])
    for i in range(3):
        rhs[3 + i] = rhs[3 + i] + w[i]
    v = d["v"]
    vv, ww = sp.Matrix(v), sp.Matrix(w)
    stage = (vv.T * sp.diag(1, 1, 1, 20, 20) * vv)[0, 0] + 10.0 * (ww.T * ww)[0, 0]
    dx = sp.Matrix([x[i] - d["x_prev"][i] for i in range(8)])
    arrival = 1e-4 * (dx.T * dx)[0, 0] + 1.0 * (p[1] - d["p_est_prev"][0]) ** 2
    dd = dict(d)
    dd.update(name="rotating_masses_mhe_w", w=w, rhs=rhs, stage_cost=stage, arrival_cost=arrival, n_horizon=6,
              nl_cons_check_colloc_points=False, nl_cons=[dict(name="phi_1_ub", expr=x[0] - 1.5, ub=0.0, soft=False)],
              p_est_lb=1e-5, p_est_ub=1e-3)
    dd.update(over)
    return dd


def case_oscillating_masses_mhe(**over):
    """A discrete-time estimator (no stored run exists): the two oscillating masses of examples/oscillating_masses_discrete with a
    measurement of the two positions (with noise), process noise on all four states, horizon 8, no estimated parameter."""
    d = case_oscillating_masses()
    x, u = d["x"], d["u"]
    w = sp.symbols("w_0:4")
    v = sp.symbols("v_0:2")
    xp = sp.symbols("xprev_0:4")
    rhs = [d["rhs"][i] + w[i] for i in range(4)]
    meas = [x[0] + v[0], x[2] + v[1]]
    vv, ww = sp.Matrix(v), sp.Matrix(w)
    dx = sp.Matrix([x[i] - xp[i] for i in range(4)])
    dd = dict(d)
    dd.update(name="oscillating_masses_mhe", v=v, w=w, rhs=rhs, meas=meas, p_est=[], x_prev=xp, p_est_prev=(), tvp=(),
              stage_cost=10.0 * (vv.T * vv)[0, 0] + 5.0 * (ww.T * ww)[0, 0], arrival_cost=0.5 * (dx.T * dx)[0, 0],
              n_horizon=8, nl_cons_check_colloc_points=False, collocation_deg=0, collocation_ni=1,
              nl_cons=[dict(name="x1_ub", expr=x[1] - 3.0, ub=0.0, soft=False)])
    dd.update(over)
    return dd


def case_oscillating_masses_dae(**over):
    """/root/reference/examples/oscillating_masses_discrete_dae/template_model.py:34-75, template_mpc.py:34-74: the discrete
    masses with the successor state as algebraic variable, x+ = z, 0 = z - A x - B u."""
    d = case_oscillating_masses()
    x, u = d["x"], d["u"]
    z = sp.symbols("x_next_0 x_next_1 x_next_2 x_next_3")
    A = np.array([[0.763, 0.460, 0.115, 0.020], [-0.899, 0.763, 0.420, 0.115], [0.115, 0.020, 0.763, 0.460],
                  [0.420, 0.115, -0.899, 0.763]])
    B = np.array([0.014, 0.063, 0.221, 0.367])
    d.update(name="oscillating_masses_dae", z=z, rhs=list(z),
             alg=[z[i] - sum(A[i, j] * x[j] for j in range(4)) - B[i] * u[0] for i in range(4)], z_scaling=np.ones(4))
    d.update(over)
    return d


def case_dip(**over):
    """/root/reference/examples/double_inverted_pendulum/template_model.py:34-146, template_mpc.py:34-100 and the obstacle /
    initial state of testing/test_DIP.py:70-90: cart with two rods, accelerations as algebraic states (Euler-Lagrange)."""
    pos, th0, th1, dpos, dth0, dth1 = x = sp.symbols("pos theta_0 theta_1 dpos dtheta_0 dtheta_1")
    ddpos, ddth0, ddth1 = z = sp.symbols("ddpos ddtheta_0 ddtheta_1")
    u = (sp.Symbol("force"),)
    m1, m2 = p = sp.symbols("m1 m2")
    tvp = (sp.Symbol("pos_set"),)
    m0, L1, L2 = 0.6, 0.5, 0.5
    l1, l2 = L1 / 2, L2 / 2
    J1, J2 = (0.2 * l1 ** 2) / 3, (0.2 * l2 ** 2) / 3
    g = 9.80665
    h1 = m0 + m1 + m2
    h2 = m1 * l1 + m2 * L1
    h3 = m2 * l2
    h4 = m1 * l1 ** 2 + m2 * L1 ** 2 + J1
    h5 = m2 * l2 * L1
    h6 = m2 * l2 ** 2 + J2
    h7 = (m1 * l1 + m2 * L1) * g
    h8 = m2 * l2 * g
    alg = [h1 * ddpos + h2 * ddth0 * sp.cos(th0) + h3 * ddth1 * sp.cos(th1)
           - (h2 * dth0 ** 2 * sp.sin(th0) + h3 * dth1 ** 2 * sp.sin(th1) + u[0]),
           h2 * sp.cos(th0) * ddpos + h4 * ddth0 + h5 * sp.cos(th0 - th1) * ddth1
           - (h7 * sp.sin(th0) - h5 * dth1 ** 2 * sp.sin(th0 - th1)),
           h3 * sp.cos(th1) * ddpos + h5 * sp.cos(th0 - th1) * ddth0 + h6 * ddth1
           - (h5 * dth0 ** 2 * sp.sin(th0 - th1) + h8 * sp.sin(th1))]
    E_kin = (sp.Rational(1, 2) * m0 * dpos ** 2
             + sp.Rational(1, 2) * m1 * ((dpos + l1 * dth0 * sp.cos(th0)) ** 2 + (l1 * dth0 * sp.sin(th0)) ** 2)
             + sp.Rational(1, 2) * J1 * dth0 ** 2
             + sp.Rational(1, 2) * m2 * ((dpos + L1 * dth0 * sp.cos(th0) + l2 * dth1 * sp.cos(th1)) ** 2
                                         + (L1 * dth0 * sp.sin(th0) + l2 * dth1 * sp.sin(th1)) ** 2)
             + sp.Rational(1, 2) * J2 * dth0 ** 2)
    E_pot = m1 * g * l1 * sp.cos(th0) + m2 * g * (L1 * sp.cos(th0) + l2 * sp.cos(th1))
    ox, oy, orad = 0.0, 0.6, 0.3
    n0, n1 = (pos, 0.0), (pos + L1 * sp.sin(th0), L1 * sp.cos(th0))
    n2 = (n1[0] + L2 * sp.sin(th1), n1[1] + L2 * sp.cos(th1))
    dist = [sp.sqrt((nx_ - ox) ** 2 + (ny_ - oy) ** 2) - orad * 1.05 for nx_, ny_ in (n0, n1, n2)]
    m_var = 0.2 * np.array([1, 0.95, 1.05])
    d = _base(name="dip", x=x, u=u, z=z, p=p, tvp=tvp, rhs=[dpos, dth0, dth1, ddpos, ddth0, ddth1], alg=alg,
              lterm=-E_pot + 10 * (pos - tvp[0]) ** 2, mterm=E_kin - E_pot, rterm=np.array([0.1]),
              n_horizon=100, n_robust=0, t_step=0.04, collocation_deg=3,
              x_lb=-np.inf * np.ones(6), x_ub=np.inf * np.ones(6), u_lb=np.array([-4.0]), u_ub=np.array([4.0]),
              x_scaling=np.ones(6), u_scaling=np.ones(1), z_scaling=np.ones(3),
              x0=np.array([0.0, 0.9 * np.pi, 0.9 * np.pi, 0.0, 0.0, 0.0]), aux={},
              nl_cons=[dict(expr=-dd, ub=0.0, soft=False) for dd in dist],
              uncertainty={"m1": m_var, "m2": m_var})
    d.update(over)
    return d


CASES = {"oscillating_masses_dae": case_oscillating_masses_dae, "dip": case_dip, "rotating_masses": case_rotating_masses, "industrial_poly": case_industrial_poly, "CSTR": case_CSTR,
         "batch_reactor": case_batch_reactor, "oscillating_masses": case_oscillating_masses,
         "kinematic_bicycle": case_kinematic_bicycle, "dynamic_bicycle": case_dynamic_bicycle, "kite": case_kite}


def p_scenarios(case):
    """All parameter combinations, first keyword varies slowest, first value nominal
    (/root/reference/do_mpc/controller/_mpc.py:867, itertools.product)."""
    names = [str(s) for s in case["p"]]
    if case.get("p_values") is not None:          # explicit scenario list (set_p_fun route, _mpc.py:760-818)
        return np.asarray(case["p_values"], float).reshape(-1, len(names))
    unc = case["uncertainty"]
    if not names:
        return np.zeros((1, 0))
    if not unc:
        return np.zeros((1, len(names)))
    combos = list(itertools.product(*[unc[k] for k in unc.keys()]))
    keys = list(unc.keys())
    out = np.zeros((len(combos), len(names)))
    for c, combo in enumerate(combos):
        for k, v in zip(keys, combo):
            out[c, names.index(k)] = v
    return out
