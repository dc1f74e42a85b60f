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
}


def p_vector(model, values: dict) -> np.ndarray:
    names = [n for n in model._p.names if n != "default"]
    return np.array([float(values[n]) for n in names])


def plant_step(model, x, u, p, t_step: float) -> np.ndarray:
    x = np.asarray(x, float).ravel()
    u = np.asarray(u, float).ravel()
    z, tvp, w = np.zeros(0), np.zeros(model.n_tvp), np.zeros(0)
    if model.model_type == "discrete":
        return np.asarray(model._rhs_fun.eval(x, u, z, tvp, p, w)[0], float).ravel()
    sol = solve_ivp(lambda t, y: np.asarray(model._rhs_fun.eval(y, u, z, tvp, p, w)[0], float).ravel(), (0.0, t_step), x,
                    method="Radau", rtol=1e-11, atol=1e-11)
    assert sol.success, sol.message
    return sol.y[:, -1]
