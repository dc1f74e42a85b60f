"""Batched sampling drivers on top of `MPC.make_step_batch`.

What they replace in the reference: the per-sample loops that feed its data-driven tools -
`do_mpc.sampling.Sampler.sample_data` (/root/reference/do_mpc/sampling/_sampler.py:198-228: one call of the user's
sample function per entry of the sampling plan, fanned out over processes by the user) and, built on it, the open-loop
sampler of the approximate-MPC module (/root/reference/do_mpc/approximateMPC/_ampc_sampler.py:234-345: draw (x0, u_prev)
uniformly in a box, `mpc.reset_history(); mpc.x0 = x0; mpc.u0 = u_prev; mpc.set_initial_guess(); u0 = mpc.make_step(x0)` per
sample, collect u0 and the solver statistics into a table).  Here the whole plan is ONE device call: every sample is an
independent cold solve with the documented initial guess, i.e. exactly what `make_step_batch` does.
"""
import time
from typing import Dict, Optional

import numpy as np


def sampling_plan_box(lbx, ubx, lbu, ubu, n_samples: int, seed: Optional[int] = None) -> Dict[str, np.ndarray]:
    """(x0, u_prev) drawn uniformly in the box [lbx, ubx] x [lbu, ubu] (_ampc_sampler.py:234-273, `gen_x0` / `gen_u_prev`),
    as arrays instead of a list of per-sample dicts; `id` numbers the samples like the reference's planner."""
    rng = np.random.default_rng(seed)
    lbx, ubx, lbu, ubu = (np.asarray(v, float).ravel() for v in (lbx, ubx, lbu, ubu))
    return {"id": np.arange(n_samples),
            "x0": rng.uniform(lbx, ubx, size=(n_samples, lbx.size)),
            "u_prev": rng.uniform(lbu, ubu, size=(n_samples, lbu.size))}


def open_loop_samples(mpc, plan: Dict[str, np.ndarray], chunk: Optional[int] = None) -> Dict[str, np.ndarray]:
    """One cold `make_step` per entry of the plan, all entries of a chunk in one launch.  Returns the columns of the
    reference's result table (_ampc_sampler.py:322-345): x0, u_prev, u0, status (= solver success), iter_count,
    t_wall (kernel time of the launch divided by its samples), t_make_step (host wall time likewise)."""
    X0 = np.asarray(plan["x0"], float)
    UP = np.asarray(plan["u_prev"], float)
    n = X0.shape[0]
    chunk = n if not chunk else int(chunk)
    nu = mpc.model.n_u
    out = {"id": np.asarray(plan.get("id", np.arange(n))), "x0": X0, "u_prev": UP, "u0": np.zeros((n, nu)),
           "status": np.zeros(n, bool), "iter_count": np.zeros(n, int), "t_wall": np.zeros(n), "t_make_step": np.zeros(n)}
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        t0 = time.perf_counter()
        r = mpc.make_step_batch(X0[lo:hi], U_prev=UP[lo:hi])
        dt = time.perf_counter() - t0
        st = r["stats"]
        out["u0"][lo:hi] = r["u0"]
        out["status"][lo:hi] = st["success"] != 0
        out["iter_count"][lo:hi] = st["iter_count"]
        out["t_wall"][lo:hi] = float(np.max(st["t_wall_total"])) / (hi - lo)
        out["t_make_step"][lo:hi] = dt / (hi - lo)
    return out


def to_dataframe(samples: Dict[str, np.ndarray]):
    """The reference's `data_<name>_all.pkl` layout (one row per sample, array-valued cells for x0 / u_prev / u0)."""
    import pandas as pd
    n = len(samples["status"])
    return pd.DataFrame({k: ([v[i] for i in range(n)] if np.ndim(v) > 1 else v) for k, v in samples.items()})


def closed_loop_samples(mpc, simulator, plan: Dict[str, np.ndarray], trajectory_length: int, device: int = 0) -> Dict[str, np.ndarray]:
    """Closed-loop sampling of the approximate-MPC module (_ampc_sampler.py:384-470: per sample `trajectory_length` steps of
    mpc.make_step -> simulator.make_step -> state feedback from (x0, u_prev), stopped at the first failed solve) for the whole
    plan at once: controller and plant advance all samples together on the GPU (`BatchClosedLoop`).  Returns per sample the
    state trajectory x[T+1], the applied inputs u[T], the previous inputs u_prev[T] (the regression features of the
    reference: `u_prev_total`), the per-step solver success and `n_valid` = number of steps before the first failure."""
    from .closed_loop import BatchClosedLoop
    X0 = np.asarray(plan["x0"], float)
    UP = np.asarray(plan["u_prev"], float)
    n, T = X0.shape[0], int(trajectory_length)
    loop = BatchClosedLoop(mpc, simulator, X0, device=device, U_prev0=UP)
    x = np.zeros((n, T + 1, X0.shape[1]))
    u = np.zeros((n, T, UP.shape[1]))
    up = np.zeros((n, T, UP.shape[1]))
    ok = np.zeros((n, T), bool)
    x[:, 0], cur_up = X0, UP.copy()
    for k in range(T):
        r = loop.step()
        up[:, k] = cur_up
        u[:, k], x[:, k + 1] = r["u0"], r["x"]
        ok[:, k] = (r["stats"]["success"] != 0) & (r["plant_status"] == 0)
        cur_up = r["u0"]
    n_valid = np.where(ok.all(axis=1), T, np.argmin(ok, axis=1))
    return {"id": np.asarray(plan.get("id", np.arange(n))), "x": x, "u": u, "u_prev": up, "success": ok, "n_valid": n_valid}
