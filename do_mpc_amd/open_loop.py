"""`open_loop=True` with several scenarios (/root/reference/do_mpc/controller/_mpc.py:1112-1117, 1205-1206): every scenario of a
stage applies the SAME input - `_u` has one scenario slot.

That NLP is not tree-structured: the shared inputs couple all scenario chains at every stage.  It is, however, a CHAIN over the
stacked scenario states: stage k of leaf scenario s evolves with x^s_{k+1} = F(x^s_k, u_k, p_{real(k, s)}), all S copies driven by the
one input sequence.  This module builds that chain problem - model with the S state copies, the parameter realisations as
time-varying parameters of the stage, cost (1/S) sum_s l(x^s, u) (the reference weights an edge of stage k with 1 / n_scenarios[k + 1]
and S / n_scenarios[k + 1] leaf scenarios pass through it) - on the controller's own machinery (MPC on the stacked model, n_robust = 0)
and maps variables, bounds, rows and multipliers between the reference's layout (`opt_x = [_x (N+1, S, 1+M) | _u (N, 1) | _eps (N, S)]`,
rows per tree edge) and the chain.  A tree node that several leaf scenarios pass through has one copy per leaf scenario; the copies
carry identical values (same initial state, inputs and parameter realisations), the reference's variable is one of them, its
multipliers the sum over them.

Cost: the collocation block of an interval has S * M * n_x unknowns (the kernels eliminate at most 64 per interval) and the Riccati
recursion runs on S * n_x states - small problems only; larger ones are refused by name.
"""
from __future__ import annotations

import numpy as np

from . import sym
from .model import Model


def check_supported(mpc) -> None:
    """refusals of the stacked formulation, by name (called from MPC.prepare_nlp)"""
    s, m, ps = mpc.settings, mpc.model, mpc.structure
    if m.n_z:
        raise NotImplementedError("structured HIP backend: open_loop with several scenarios for models with algebraic states")
    if mpc.rterm_expr is not None:
        raise NotImplementedError("structured HIP backend: open_loop with several scenarios and a user-defined rterm expression")
    if getattr(mpc, "_nl_colloc", False):
        raise NotImplementedError("structured HIP backend: open_loop with several scenarios and nl_cons_check_colloc_points")
    if ps.ns and (s.n_robust >= 2 or s.nl_cons_single_slack):
        # (n_robust >= 2: the slack of a shared node is paid once per BRANCH in the reference, _mpc.py:1254, but has one copy per leaf
        #  scenario in the stacked problem - a stage-dependent weight the generated code does not have)
        raise NotImplementedError("structured HIP backend: open_loop with several scenarios and soft constraints for n_robust >= 2 / "
                                  "nl_cons_single_slack")
    if ps.ns and m.n_p and s.n_robust >= 1 and any(sym.depends_on(sym.SX(c["expr"]).nodes(), m._p.cat.nodes()) for c in mpc.nl_cons_list):
        # (the slack of a branching node covers the LARGEST violation over its branches in the reference, one slack per leaf scenario covers
        #  each branch's own: the same problem only if the rows do not depend on the uncertain parameters)
        raise NotImplementedError("structured HIP backend: open_loop with several scenarios and soft constraints whose expressions "
                                  "depend on the uncertain parameters")
    n_w = ps.S * ps.M * ps.nx
    if n_w > 64 or (ps.M == 0 and ps.S * ps.nx > 64):
        raise NotImplementedError("structured HIP backend: open_loop with {} scenarios: {} stacked collocation unknowns per control "
                                  "interval (scenarios * (deg + 1) * ni * n_x); the kernels eliminate at most 64".format(ps.S, n_w))


class OpenLoopStack:
    """Stands where the solver object stands (`MPC.S`): `r = S(x0=, lbx=, ubx=, lbg=, ubg=, p=)`, `S.stats()`, `S.solve_batch(...)` in the
    reference's layouts; inside, the chain problem over the stacked scenario states on a HipIpmSolver."""

    shard_capable = False

    def __init__(self, mpc, solver_factory=None):
        from .controller import MPC
        s, m, ps = mpc.settings, mpc.model, mpc.structure
        S, nx, nu, npar, ntvp, N, M = ps.S, ps.nx, ps.nu, ps.np_, ps.ntvp, ps.N, ps.M
        self.ps = ps
        # ---- the stacked model: S copies of the states, shared inputs, the parameter realisations as time-varying parameters
        sm = Model(m.model_type, m.symvar_type)
        xs = [sm.set_variable("_x", "x_s%d" % c, (nx, 1)) for c in range(S)]
        for n in m._u.names:
            if m._u.vars[n].numel():
                sm._u.add(n, m._u.vars[n])
        for n in m._tvp.names:
            if m._tvp.vars[n].numel():
                sm._tvp.add(n, m._tvp.vars[n])
        pc = [sm.set_variable("_tvp", "p_s%d" % c, (npar, 1)) for c in range(S)] if npar else [sym.SX([], (0, 1))] * S
        sp = sym.SX(np.asarray(mpc._p_scaling.master, float).reshape(-1, 1)) if npar else None
        rhs = m._rhs
        if m.n_w and sym.depends_on(rhs.nodes(), m._w.cat.nodes()):
            rhs = sym.substitute(rhs, m._w.cat, sym.SX.zeros(m.n_w, 1))          # _w = 0 in the MPC (_mpc.py:1166)

        def at(expr, c, scaled_p):
            e = sym.SX(expr)
            if npar:      # (the reference multiplies the parameters by `_p_scaling` in the model equations only, optimizer.py:808-812)
                e = sym.substitute(e, m._p.cat, pc[c] * sp if scaled_p else pc[c])
            return sym.substitute(e, m._x.cat, xs[c])

        for c in range(S):
            sm.set_rhs("x_s%d" % c, at(rhs, c, True))
        sm.setup()
        self.stacked_model = sm
        inner = MPC(sm)
        st = inner.settings
        st.n_horizon, st.t_step, st.n_robust, st.open_loop = N, s.t_step, 0, False
        st.state_discretization, st.collocation_type = s.state_discretization, s.collocation_type
        st.collocation_deg, st.collocation_ni = s.collocation_deg, s.collocation_ni
        st.nlpsol_opts = dict(s.nlpsol_opts)
        st.gpu_index, st.max_batch, st.block_threads = s.gpu_index, s.max_batch, s.block_threads
        w = 1.0 / S
        inner.set_objective(mterm=sum((w * at(mpc.mterm, c, False) for c in range(S)), sym.SX(0.0)),
                            lterm=sum((w * at(mpc.lterm, c, False) for c in range(S)), sym.SX(0.0)))
        inner.set_rterm()
        inner.rterm_factor.master[:] = mpc.rterm_factor.master
        inner.flags["set_rterm"] = True
        soft = {sl["slack_name"]: sl for sl in mpc.slack_vars_list}
        for c in range(S):                                  # rows of a stage: copy-major, the user's order inside a copy
            for con in mpc.nl_cons_list:
                sl = soft.get(con["expr_name"])
                inner.set_nl_cons("%s_s%d" % (con["expr_name"], c), at(con["expr"], c, False), ub=con["ub"], soft_constraint=sl is not None,
                                  penalty_term_cons=sl["penalty"] if sl else 1, maximum_violation=sl["ub"] if sl else np.inf)
        inner._x_scaling.master[:] = np.tile(mpc._x_scaling.master, S)
        inner._u_scaling.master[:] = mpc._u_scaling.master
        inner.set_tvp_fun(lambda t: inner.get_tvp_template())
        inner.prepare_nlp()
        inner.create_nlp(_solver_factory=solver_factory)
        self.inner = inner
        ips = self.ips = inner.structure
        assert ips.nx == S * nx and ips.nu == nu and ips.ns == S * ps.ns and ips.ne == S * ps.ne and ips.ntvp == ntvp + S * npar
        # ---- maps chain -> reference
        T = ps.tables
        n_scen = ps.scenario_tree["n_scenarios"]
        lvl = T["level_node_start"]
        anc = lambda k, c: c // (S // n_scen[k])           # noqa: E731   node index at level k of leaf scenario c's ancestor
        xmap = -np.ones(ips.n_opt_x, dtype=np.int64)
        gmap = -np.ones(ips.n_g, dtype=np.int64)
        a_x, a_u, a_s = np.arange(nx), np.arange(nu), np.arange(ps.ns)
        real = np.zeros((N + 1, S), dtype=np.int64)         # parameter realisation of copy c at stage k (stage N: the last edge's, _mpc.py:1257-1259)
        for c in range(S):
            gmap[c * nx + a_x] = a_x                        # initial-condition rows (every copy of the root)
            for k in range(N + 1):
                for slot in range(M + 1):
                    xmap[ips.ix(k, 0, slot) + c * nx + a_x] = ps.ix(k, anc(k, c), slot) + a_x
            for k in range(N):
                if ps.ns:
                    xmap[ips.ieps(k, 0) + c * ps.ns + a_s] = ps.ieps(k, anc(k, c)) + a_s
                e = int(T["node_in_edge"][lvl[k + 1] + anc(k + 1, c)])       # the tree edge of stage k on leaf scenario c's path
                real[k, c] = T["edge_pidx"][e]
                r_ref, r_in = int(T["edge_row0"][e]), int(ips.tables["edge_row0"][k])
                for blk in range(M):
                    gmap[r_in + blk * ips.nx + c * nx + a_x] = r_ref + blk * nx + a_x
                gmap[r_in + M * ips.nx + c * nx + a_x] = r_ref + M * nx + a_x
                if ps.ne:
                    a_e = np.arange(ps.ne)
                    gmap[r_in + M * ips.nx + ips.nx + c * ps.ne + a_e] = r_ref + M * nx + nx + a_e
            real[N, c] = real[N - 1, c]
        for k in range(N):
            xmap[ips.iu(k, 0) + a_u] = ps.iu(k, 0) + a_u
        assert xmap.min() >= 0 and gmap.min() >= 0
        self.xmap, self.gmap, self.real = xmap, gmap, real
        self.options = getattr(inner.S, "options", None)
        self.ignored_options = getattr(inner.S, "ignored_options", [])

    # ------------------------------------------------------------------ layouts
    def _p_in(self, P: np.ndarray) -> np.ndarray:
        """[_x0 | _tvp (N+1) | _p (n_comb) | _u_prev] -> [_x0 of every copy | (_tvp_k, p_real(k, c) for every copy) per stage | _u_prev]"""
        ps, ips = self.ps, self.ips
        P = np.asarray(P, dtype=float)
        lead = P.shape[:-1]
        out = np.zeros(lead + (ips.n_opt_p,))
        out[..., :ips.nx] = np.tile(P[..., :ps.nx], ps.S)
        TV = out[..., ips.p_off_tvp:ips.p_off_p].reshape(lead + (ps.N + 1, ips.ntvp))
        if ps.ntvp:
            TV[..., :ps.ntvp] = P[..., ps.p_off_tvp:ps.p_off_p].reshape(lead + (ps.N + 1, ps.ntvp))
        if ps.np_:
            Pm = P[..., ps.p_off_p:ps.p_off_uprev].reshape(lead + (ps.n_comb, ps.np_))
            TV[..., ps.ntvp:] = Pm[..., self.real, :].reshape(lead + (ps.N + 1, ps.S * ps.np_))
        out[..., ips.p_off_uprev:] = P[..., ps.p_off_uprev:]
        return out

    def _out(self, r: dict, x_in: np.ndarray) -> dict:
        ps = self.ps
        x_i = np.asarray(r["x"], float)
        lead = x_i.shape[:-1]
        x = np.array(np.broadcast_to(x_in, lead + (ps.n_opt_x,)), dtype=float)      # (variables no node reads keep the caller's values)
        x[..., self.xmap] = x_i
        g = np.zeros(lead + (ps.n_g,))
        g[..., self.gmap] = np.asarray(r["g"], float)
        lam_g, lam_x = np.zeros(lead + (ps.n_g,)), np.zeros(lead + (ps.n_opt_x,))
        if lead:
            for b in range(int(np.prod(lead))):
                np.add.at(lam_g.reshape(-1, ps.n_g)[b], self.gmap, np.asarray(r["lam_g"], float).reshape(-1, self.gmap.size)[b])
                np.add.at(lam_x.reshape(-1, ps.n_opt_x)[b], self.xmap, np.asarray(r["lam_x"], float).reshape(-1, self.xmap.size)[b])
        else:
            np.add.at(lam_g, self.gmap, np.asarray(r["lam_g"], float))
            np.add.at(lam_x, self.xmap, np.asarray(r["lam_x"], float))
        out = dict(r)
        out.update(x=x, g=g, lam_g=lam_g, lam_x=lam_x)
        return out

    # ------------------------------------------------------------------ the solver object's surface
    def __call__(self, x0, lbx, ubx, lbg, ubg, p, lam_x0=None, lam_g0=None):
        x0, lbx, ubx = (np.asarray(a, float).reshape(-1) for a in (x0, lbx, ubx))
        lbg, ubg = np.asarray(lbg, float).reshape(-1), np.asarray(ubg, float).reshape(-1)
        r = self.inner.S(x0=x0[self.xmap], lbx=lbx[self.xmap], ubx=ubx[self.xmap], lbg=lbg[self.gmap], ubg=ubg[self.gmap],
                         p=self._p_in(np.asarray(p, float).reshape(-1)))
        return self._out(r, x0)

    def solve_batch(self, Xi, lbx, ubx, lbg, ubg, P):
        Xi = np.asarray(Xi, float)
        lbx, ubx, lbg, ubg = (np.asarray(a, float).reshape(-1) for a in (lbx, ubx, lbg, ubg))
        r = self.inner.S.solve_batch(Xi[:, self.xmap], lbx[self.xmap], ubx[self.xmap], lbg[self.gmap], ubg[self.gmap], self._p_in(P))
        return self._out(r, Xi)

    def stats(self):
        return self.inner.S.stats()

    def close(self):
        self.inner.S.close()
