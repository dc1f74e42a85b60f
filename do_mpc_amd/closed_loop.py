"""Batched closed loop with the whole x0 batch resident in HBM.

The reference closes the loop one sample at a time on the host - `u0 = mpc.make_step(x0)`, `y = simulator.make_step(u0)`,
`x0 = estimator.make_step(y)` (/root/reference/examples/industrial_poly/main.py:105-108) - and fans batches of such loops
out over processes (`do_mpc.sampling.Sampler`, /root/reference/do_mpc/sampling/_sampler.py:198-228).  Here B loops
advance together: one `dompc_solve_batch_device` launch (structured IPM, B problems) and one
`dompc_plant_step_batch_device` launch (plant integrator, B samples) per control step on the same stream; states,
inputs, parameters and the warm-start solution stay in device memory (torch tensors are only the allocator).
State feedback (`StateFeedback.make_step` returns y, estimator/_base.py:63-72), measurement = state.
"""
from __future__ import annotations

import numpy as np

from .solver import STATS_DTYPE


class BatchClosedLoop:
    def __init__(self, mpc, simulator, X0, device: int = 0, U_prev0=None):
        import torch
        self.torch = torch
        self.mpc, self.sim = mpc, simulator
        ps = self.ps = mpc.structure
        if getattr(ps, "open_loop_stack", False):
            raise NotImplementedError("structured HIP backend: the device-resident closed loop with open_loop and several scenarios "
                                      "(the controller's vectors live in the stacked chain layout there; use make_step_batch)")
        if getattr(mpc.S, "row_mapped", False):
            raise NotImplementedError("structured HIP backend: the device-resident closed loop with rows appended to nlp_cons (internal row "
                                      "layout, solver.RowMappedSolver); use make_step_batch")
        m = simulator.model
        assert m.n_x == ps.nx and m.n_u == ps.nu, "controller and plant must share states and inputs"
        assert m.n_y == m.n_x, "state feedback: the plant's measurement must be its state"
        X0 = np.asarray(X0, dtype=float).reshape(-1, ps.nx)
        self.B = B = X0.shape[0]
        dev = self.dev = torch.device("cuda", device)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)      # noqa: E731
        t0 = float(mpc._t0[0])
        P = np.tile(mpc.opt_p_num.master, (B, 1))
        P[:, :ps.nx] = X0
        P[:, ps.p_off_tvp:ps.p_off_p] = mpc.tvp_fun(t0).master
        P[:, ps.p_off_p:ps.p_off_uprev] = mpc.p_fun(t0).master
        P[:, ps.p_off_uprev:] = 0.0 if U_prev0 is None else np.asarray(U_prev0, float).reshape(B, ps.nu)
        Xi = np.zeros((B, ps.n_opt_x))
        Xi[:, :ps.off_z].reshape(B, -1, ps.nx)[:] = (X0 / mpc._x_scaling.master)[:, None, :]
        Xi[:, ps.off_u:ps.off_eps].reshape(B, -1, ps.nu)[:] = (P[:, ps.p_off_uprev:] / mpc._u_scaling.master)[:, None, :]   # (mpc.u0 = u_prev; set_initial_guess)
        self.P, self.guess = t(P), t(Xi)                      # opt_p per sample; initial guess = set_initial_guess semantics
        self.X = t(X0)                                        # plant states
        self.lbx, self.ubx = t(mpc._lb_opt_x.master), t(mpc._ub_opt_x.master)
        self.lbg, self.ubg = t(mpc._nlp_cons_lb), t(mpc._nlp_cons_ub)
        self.sol = torch.empty((B, ps.n_opt_x), dtype=torch.float64, device=dev)
        self.f = torch.empty(B, dtype=torch.float64, device=dev)
        self.stats = torch.zeros(B * STATS_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        self.U = torch.empty((B, ps.nu), dtype=torch.float64, device=dev)
        self.Xn = torch.empty_like(self.X)
        self.pstat = torch.zeros(B, dtype=torch.int32, device=dev)
        self.xs, self.us = t(mpc._x_scaling.master), t(mpc._u_scaling.master)
        ts = float(simulator._t0[0])
        self.p_plant = t(simulator.p_fun(ts).master if m.n_p else np.zeros(1))
        self.tvp_plant = t(simulator.tvp_fun(ts).master if m.n_tvp else np.zeros(1))
        self.k = 0
        # loop time: the reference's per-sample loop re-evaluates tvp_fun / p_fun at the current time in EVERY make_step of
        # controller and plant (_mpc.py:1009-1019, simulator.py:790-800) - a set-point staircase over the horizon moves along
        self.t_mpc0, self.t_sim0 = t0, ts
        self.dt_mpc, self.dt_sim = float(mpc.settings.t_step), float(simulator.settings.t_step)
        self._p_row = P[0].copy()                             # host image of one opt_p row (its _tvp / _p blocks are re-uploaded)

    def _refresh_time_varying(self):
        """_tvp / _p of controller and plant at the current loop time -> device (same values for every sample of the batch)"""
        ps, m, torch = self.ps, self.sim.model, self.torch
        tm, ts = self.t_mpc0 + self.k * self.dt_mpc, self.t_sim0 + self.k * self.dt_sim
        if ps.ntvp or ps.np_:
            row = self._p_row
            row[ps.p_off_tvp:ps.p_off_p] = self.mpc.tvp_fun(tm).master
            row[ps.p_off_p:ps.p_off_uprev] = self.mpc.p_fun(tm).master
            blk = torch.from_numpy(row[ps.p_off_tvp:ps.p_off_uprev].copy()).to(self.dev)
            self.P[:, ps.p_off_tvp:ps.p_off_uprev] = blk
        if m.n_p:
            self.p_plant.copy_(torch.from_numpy(np.ascontiguousarray(self.sim.p_fun(ts).master, dtype=np.float64)).to(self.dev))
        if m.n_tvp:
            self.tvp_plant.copy_(torch.from_numpy(np.ascontiguousarray(self.sim.tvp_fun(ts).master, dtype=np.float64)).to(self.dev))

    def step(self) -> dict:
        """one control step of all B loops; returns the solver statistics (numpy record array) and the plant status"""
        torch, ps, S = self.torch, self.ps, self.mpc.S
        stream = torch.cuda.current_stream()
        if self.k > 0:
            self._refresh_time_varying()
        S.solve_batch_device(self.B, self.guess.data_ptr(), self.lbx.data_ptr(), self.ubx.data_ptr(), self.lbg.data_ptr(),
                             self.ubg.data_ptr(), self.P.data_ptr(), self.sol.data_ptr(), 0, 0, 0, self.f.data_ptr(),
                             self.stats.data_ptr(), stream=stream.cuda_stream)
        iu = ps.iu(0, 0)
        torch.mul(self.sol[:, iu:iu + ps.nu], self.us, out=self.U)                      # u0 in physical units
        self.sim.step_batch_device(self.B, self.X.data_ptr(), self.U.data_ptr(), self.tvp_plant.data_ptr(),
                                   self.p_plant.data_ptr(), self.Xn.data_ptr(), 0, self.pstat.data_ptr(),
                                   shared_mask=2 | 4 | 8 | 16, stream=stream.cuda_stream)
        # next problem: x0 <- plant state, u_prev <- applied input, initial guess <- previous solution (optimizer.py:754-768)
        self.X, self.Xn = self.Xn, self.X
        self.P[:, :ps.nx] = self.X
        self.P[:, ps.p_off_uprev:] = self.U
        self.guess.copy_(self.sol)
        self.k += 1
        torch.cuda.synchronize()
        st = np.frombuffer(self.stats.cpu().numpy().tobytes(), dtype=STATS_DTYPE).copy()
        return {"stats": st, "plant_status": (self.pstat.cpu().numpy() & 1), "plant_implicit": ((self.pstat.cpu().numpy() >> 1) & 1), "u0": self.U.cpu().numpy(), "x": self.X.cpu().numpy()}


class BatchClosedLoopMHE:
    """B closed loops controller -> plant -> moving horizon estimator -> controller advancing together, everything resident in HBM
    (the loop of examples/rotating_oscillating_masses_mhe_mpc/main.py, one sample at a time on the host there): per control step
    ONE batched solve of the controller (B problems), ONE batched plant step (states and measurements), ONE batched solve of the
    estimator (B estimation problems), plus index arithmetic on device tensors - the estimator's parameter vectors (previous
    estimates, sliding measurement window) and initial guesses are assembled from the previous device results.
    The controller's and the estimator's problems live in their chain layouts (do_mpc_amd/estimator.py); torch is the allocator
    and does the slicing."""

    def __init__(self, mpc, simulator, mhe, X0_true, x0_est=None, p_est0=None, device: int = 0):
        import torch
        self.torch = torch
        self.mpc, self.sim, self.mhe = mpc, simulator, mhe
        m = simulator.model
        ps, es = mpc.structure, mhe._ps                     # controller / estimator chain structures
        if getattr(ps, "open_loop_stack", False):
            raise NotImplementedError("structured HIP backend: the device-resident closed loop with open_loop and several scenarios")
        if getattr(mpc.S, "row_mapped", False):
            raise NotImplementedError("structured HIP backend: the device-resident closed loop with rows appended to nlp_cons")
        self.ps, self.es = ps, es
        X0_true = np.asarray(X0_true, dtype=float).reshape(-1, m.n_x)
        self.B = B = X0_true.shape[0]
        dev = self.dev = torch.device("cuda", device)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)      # noqa: E731
        nx, nu, ny, npe = m.n_x, m.n_u, m.n_y, mhe.n_p_est
        self.nx, self.nu, self.ny, self.npe = nx, nu, ny, npe
        x_est = np.zeros((B, nx)) if x0_est is None else np.broadcast_to(np.asarray(x0_est, float).reshape(-1, nx), (B, nx)).copy()
        p_est = np.zeros((B, npe)) if p_est0 is None else np.broadcast_to(np.asarray(p_est0, float).reshape(-1, npe), (B, npe)).copy()
        # ---- controller: opt_p rows, initial guess (set_initial_guess semantics), bounds
        t0 = float(mpc._t0[0])
        Pc = np.tile(mpc.opt_p_num.master, (B, 1))
        Pc[:, :nx] = x_est
        Pc[:, ps.p_off_tvp:ps.p_off_p] = mpc.tvp_fun(t0).master
        Pc[:, ps.p_off_p:ps.p_off_uprev] = mpc.p_fun(t0).master
        Pc[:, ps.p_off_uprev:] = 0.0
        Gc = np.zeros((B, ps.n_opt_x))
        Gc[:, :ps.off_z].reshape(B, -1, ps.nx)[:] = (x_est / mpc._x_scaling.master)[:, None, :]
        self.Pc, self.Gc = t(Pc), t(Gc)
        self.c_lbx, self.c_ubx = t(mpc._lb_opt_x.master), t(mpc._ub_opt_x.master)
        self.c_lbg, self.c_ubg = t(mpc._nlp_cons_lb), t(mpc._nlp_cons_ub)
        self.c_sol = torch.empty((B, ps.n_opt_x), dtype=torch.float64, device=dev)
        self.c_f = torch.empty(B, dtype=torch.float64, device=dev)
        self.c_stats = torch.zeros(B * STATS_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        self.xs, self.us = t(mpc._x_scaling.master), t(mpc._u_scaling.master)
        # ---- plant
        self.X = t(X0_true)
        self.Xn = torch.empty_like(self.X)
        self.U = torch.empty((B, nu), dtype=torch.float64, device=dev)
        self.Y = torch.empty((B, ny), dtype=torch.float64, device=dev)
        self.pstat = torch.zeros(B, dtype=torch.int32, device=dev)
        ts = float(simulator._t0[0])
        self.p_plant = t(simulator.p_fun(ts).master if m.n_p else np.zeros(1))
        self.tvp_plant = t(simulator.tvp_fun(ts).master if m.n_tvp else np.zeros(1))
        # ---- estimator: chain opt_p rows (previous estimate | (tvp_k, y_k) per stage | p_set | 0), initial guess, bounds
        N = self.N = mhe.settings.n_horizon
        te = float(mhe._t0[0])
        op = np.zeros((B, mhe.n_opt_p))
        op[:, :nx] = x_est
        op[:, nx:nx + npe] = p_est
        op[:, mhe._po_pset:mhe._po_tvp] = mhe.p_fun(te).master
        op[:, mhe._po_tvp:mhe._po_y] = mhe.tvp_fun(te).master
        ge = np.zeros((B, mhe.n_opt_x))
        ge[:, :mhe._o_z].reshape(B, -1, nx)[:] = (x_est / mhe._x_scaling.master)[:, None, :]
        pes = mhe._p_est_scaling.master if npe else np.zeros(0)
        ge[:, mhe._o_p:] = p_est / pes if npe else p_est      # (opt_x holds the SCALED parameter, _mhe.py:939-941 / set_initial_guess)
        self.Pe, self.Ge = t(mhe._p_to_chain(op)), t(mhe._to_chain(ge))
        em = mhe._mpc
        self.e_lbx, self.e_ubx = t(em._lb_opt_x.master), t(em._ub_opt_x.master)
        self.e_lbg, self.e_ubg = t(em._nlp_cons_lb), t(em._nlp_cons_ub)
        self.e_sol = torch.empty((B, es.n_opt_x), dtype=torch.float64, device=dev)
        self.e_f = torch.empty(B, dtype=torch.float64, device=dev)
        self.e_stats = torch.zeros(B * STATS_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        self.x_est, self.p_est = t(x_est), t(p_est)
        self.exs = t(mhe._x_scaling.master)
        self.eps_ = t(pes if npe else np.zeros(1))              # scaling of the estimated parameters (they ride as scaled states of the chain problem)
        self.t_mpc0, self.dt = t0, float(mpc.settings.t_step)
        self.t_sim0, self.dt_sim = ts, float(simulator.settings.t_step)
        self.t_mhe0, self.dt_mhe = te, float(mhe.settings.t_step)
        self._pc_row = Pc[0].copy()
        self.k = 0

    def _refresh_time_varying(self):
        """_tvp / _p of controller, plant AND estimator at the current loop time -> device (same values for every sample).  The
        reference's per-sample loop re-evaluates all of them in every make_step (_mpc.py:1009-1019, simulator.py:790-800,
        _mhe.py:940-950); only the measurement columns of the estimator's window are data of the loop."""
        torch, ps, es, m, mhe = self.torch, self.ps, self.es, self.sim.model, self.mhe
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.dev)      # noqa: E731
        tm, ts, te = self.t_mpc0 + self.k * self.dt, self.t_sim0 + self.k * self.dt_sim, self.t_mhe0 + self.k * self.dt_mhe
        if ps.ntvp or ps.np_:
            row = self._pc_row
            row[ps.p_off_tvp:ps.p_off_p] = self.mpc.tvp_fun(tm).master
            row[ps.p_off_p:ps.p_off_uprev] = self.mpc.p_fun(tm).master
            self.Pc[:, ps.p_off_tvp:ps.p_off_uprev] = up(row[ps.p_off_tvp:ps.p_off_uprev].copy())
        if m.n_p:
            self.p_plant.copy_(up(self.sim.p_fun(ts).master))
        if m.n_tvp:
            self.tvp_plant.copy_(up(self.sim.tvp_fun(ts).master))
        mt = mhe.model.n_tvp
        if mt or mhe.n_p_set:
            op = np.zeros((1, mhe.n_opt_p))
            op[:, mhe._po_pset:mhe._po_tvp] = mhe.p_fun(te).master
            op[:, mhe._po_tvp:mhe._po_y] = mhe.tvp_fun(te).master
            pc = mhe._p_to_chain(op)[0]
            if mt:
                tv = pc[es.p_off_tvp:es.p_off_p].reshape(self.N + 1, es.ntvp)[:, :mt]
                self.Pe[:, es.p_off_tvp:es.p_off_p].view(self.B, self.N + 1, es.ntvp)[:, :, :mt] = up(tv.copy())
            if es.p_off_uprev > es.p_off_p:
                self.Pe[:, es.p_off_p:es.p_off_uprev] = up(pc[es.p_off_p:es.p_off_uprev].copy())

    def step(self) -> dict:
        torch, ps, es, B, nx, nu, ny, npe, N = self.torch, self.ps, self.es, self.B, self.nx, self.nu, self.ny, self.npe, self.N
        stream = torch.cuda.current_stream()
        mt = self.mhe.model.n_tvp
        if self.k > 0:
            self._refresh_time_varying()
        # 1. controller
        self.mpc.S.solve_batch_device(B, self.Gc.data_ptr(), self.c_lbx.data_ptr(), self.c_ubx.data_ptr(), self.c_lbg.data_ptr(),
                                      self.c_ubg.data_ptr(), self.Pc.data_ptr(), self.c_sol.data_ptr(), 0, 0, 0, self.c_f.data_ptr(),
                                      self.c_stats.data_ptr(), stream=stream.cuda_stream)
        iu = ps.iu(0, 0)
        torch.mul(self.c_sol[:, iu:iu + nu], self.us, out=self.U)
        # 2. plant: next true state and its measurement
        self.sim.step_batch_device(B, self.X.data_ptr(), self.U.data_ptr(), self.tvp_plant.data_ptr(), self.p_plant.data_ptr(),
                                   self.Xn.data_ptr(), self.Y.data_ptr(), self.pstat.data_ptr(), shared_mask=2 | 4 | 8 | 16,
                                   stream=stream.cuda_stream)
        self.X, self.Xn = self.Xn, self.X
        # 3. estimator: previous estimates <- `_x[1, -1]`, `_p_est` of ITS previous solution (initial guess before the first solve,
        #    _mhe.py:939-941), measurement window shifted by one stage, the new measurement last
        i1 = es.ix(1, 0, es.M)
        self.Pe[:, :nx] = self.Ge[:, i1:i1 + nx] * self.exs
        self.Pe[:, nx:nx + npe] = self.p_est
        TV = self.Pe[:, es.p_off_tvp:es.p_off_p].view(B, N + 1, es.ntvp)
        TV[:, :N - 1, mt:] = TV[:, 1:N, mt:].clone()
        TV[:, N - 1, mt:] = self.Y
        self.mhe.S.solve_batch_device(B, self.Ge.data_ptr(), self.e_lbx.data_ptr(), self.e_ubx.data_ptr(), self.e_lbg.data_ptr(),
                                      self.e_ubg.data_ptr(), self.Pe.data_ptr(), self.e_sol.data_ptr(), 0, 0, 0, self.e_f.data_ptr(),
                                      self.e_stats.data_ptr(), stream=stream.cuda_stream)
        iN = es.ix(N, 0, es.M)
        self.x_est = self.e_sol[:, iN:iN + nx] * self.exs
        i0 = es.ix(0, 0, es.M)
        # (physical units: the chain problem carries the parameter as a SCALED state; `_p_est_prev` of the arrival cost and the
        #  returned estimate are physical, _mhe.py: opt_x_num['_p_est'] * _p_est_scaling - ADVICE r3)
        self.p_est = self.e_sol[:, i0 + nx:i0 + nx + npe] * self.eps_[:npe] if npe else self.e_sol[:, i0 + nx:i0 + nx].clone()
        self.Ge.copy_(self.e_sol)                                  # warm start (unshifted, optimizer.py:754-768)
        # 4. next controller problem: x0 <- estimate, u_prev <- applied input, initial guess <- previous solution
        self.Pc[:, :nx] = self.x_est
        self.Pc[:, ps.p_off_uprev:] = self.U
        self.Gc.copy_(self.c_sol)
        self.k += 1
        torch.cuda.synchronize()
        cs = np.frombuffer(self.c_stats.cpu().numpy().tobytes(), dtype=STATS_DTYPE).copy()
        est = np.frombuffer(self.e_stats.cpu().numpy().tobytes(), dtype=STATS_DTYPE).copy()
        return {"mpc_stats": cs, "mhe_stats": est, "plant_status": (self.pstat.cpu().numpy() & 1), "plant_implicit": ((self.pstat.cpu().numpy() >> 1) & 1), "u0": self.U.cpu().numpy(),
                "x_true": self.X.cpu().numpy(), "y": self.Y.cpu().numpy(), "x_est": self.x_est.cpu().numpy(), "p_est": self.p_est.cpu().numpy()}
