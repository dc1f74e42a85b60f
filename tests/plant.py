"""TEST-ONLY plant integrator: stands in for do_mpc.simulator.Simulator (CVODES, abstol = reltol = 1e-10,
/root/reference/examples/*/template_simulator.py) so that the reference's closed-loop tests
(/root/reference/testing/test_{CSTR,batch_reactor,industrial_poly,oscillating_masses_discrete}.py: 5 steps of
mpc.make_step -> simulator.make_step) can be repeated against their golden trajectories without CasADi/SUNDIALS.
scipy's Radau at rtol = atol = 1e-11 on the model's own right-hand side (do_mpc_amd.sym numeric evaluation)."""
import numpy as np
from scipy.integrate import solve_ivp

# true plant parameters of the reference simulators (template_simulator.py:p_fun)
PLANT_P = {
    "CSTR": {"alpha": 1.0, "beta": 1.0},
    "batch_reactor": {"Y_x": 0.5, "S_in": 200.0},
    "industrial_poly": {"delH_R": 950.0, "k_0": 7.0},
    "oscillating_masses": {},
    "oscillating_masses_dae": {},
    "dip": {"m1": 0.2, "m2": 0.2},                       # (examples/double_inverted_pendulum/template_simulator.py)
}


def p_vector(model, values: dict) -> np.ndarray:
    names = [n for n in model._p.names if n != "default"]
    return np.array([float(values[n]) for n in names])


def _solve_z(model, x, u, tvp, p, w, z0):
    """algebraic states of a semi-explicit index-1 DAE at (x, u): Newton's method on alg(x, z) = 0 with a finite-difference
    Jacobian (test-side reference: independent of the generated code)"""
    z = np.array(z0, float)
    f = lambda zz: np.asarray(model._alg_fun.eval(x, u, zz, tvp, p, w)[0], float).ravel()      # noqa: E731
    for _ in range(50):
        a = f(z)
        J = np.zeros((z.size, z.size))
        for j in range(z.size):
            h = 1e-7 * max(1.0, abs(z[j]))
            zp = z.copy(); zp[j] += h
            zm = z.copy(); zm[j] -= h
            J[:, j] = (f(zp) - f(zm)) / (2 * h)
        dz = np.linalg.solve(J, a)
        z = z - dz
        if np.max(np.abs(dz)) <= 1e-13 * max(1.0, np.max(np.abs(z))):
            break
    return z


def plant_step(model, x, u, p, t_step: float, tvp=None) -> np.ndarray:
    x = np.asarray(x, float).ravel()
    u = np.asarray(u, float).ravel()
    tvp = np.zeros(model.n_tvp) if tvp is None else np.asarray(tvp, float).ravel()
    w = np.zeros(0)
    if model.n_z:
        state = {"z": np.zeros(model.n_z)}

        def rhs(y):
            state["z"] = _solve_z(model, y, u, tvp, p, w, state["z"])
            return np.asarray(model._rhs_fun.eval(y, u, state["z"], tvp, p, w)[0], float).ravel()
        if model.model_type == "discrete":
            return rhs(x)
        sol = solve_ivp(lambda t, y: rhs(y), (0.0, t_step), x, method="Radau", rtol=1e-11, atol=1e-11)
        assert sol.success, sol.message
        return sol.y[:, -1]
    z = np.zeros(0)
    if model.model_type == "discrete":
        return np.asarray(model._rhs_fun.eval(x, u, z, tvp, p, w)[0], float).ravel()
    sol = solve_ivp(lambda t, y: np.asarray(model._rhs_fun.eval(y, u, z, tvp, p, w)[0], float).ravel(), (0.0, t_step), x,
                    method="Radau", rtol=1e-11, atol=1e-11)
    assert sol.success, sol.message
    return sol.y[:, -1]
