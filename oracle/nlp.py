"""Flat (unstructured, sparse) restatement of the NLP that do-mpc hands to nlpsol.

TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.

Follows, line by line, the reference's NLP definition:
  variables / ordering   /root/reference/do_mpc/controller/_mpc.py:1126-1134
  parameters             /root/reference/do_mpc/controller/_mpc.py:1160-1165
  constraints + cost     /root/reference/do_mpc/controller/_mpc.py:1189-1275
  collocation integrator /root/reference/do_mpc/optimizer.py:840-990
  scenario tree          /root/reference/do_mpc/optimizer.py:1011-1048
  bounds                 /root/reference/do_mpc/controller/_mpc.py:1061-1095
  slack / nl_cons        /root/reference/do_mpc/optimizer.py:543-585
Everything is kept in the reference's *scaled* variables.  Derivatives come from
sympy (oracle/models.py), assembled into scipy.sparse matrices; no structure is
exploited, which is the point: the product's Riccati solver is checked against a
solver that knows nothing about stages.
"""
import numpy as np
import scipy.sparse as sps
import sympy as sp

from .models import p_scenarios


# ----------------------------------------------------------------------------- collocation
def collocation_points(deg, kind):
    """Roots used by casadi.collocation_points (optimizer.py:844-849)."""
    if kind == "radau":
        # Radau IIA points on (0,1]: roots of P_{d-1}(2t-1) - P_d(2t-1) ... computed via
        # Jacobi polynomial P^{(1,0)}_{d-1} plus endpoint 1.
        from scipy.special import roots_jacobi
        if deg == 1:
            return [1.0]
        r, _ = roots_jacobi(deg - 1, 1.0, 0.0)
        return list((r + 1.0) / 2.0) + [1.0]
    if kind == "legendre":
        from scipy.special import roots_legendre
        r, _ = roots_legendre(deg)
        return list((r + 1.0) / 2.0)
    raise ValueError("Unknown collocation scheme")


def collocation_coeffs(deg, kind):
    """C[r,j] = d/dtau L_r (tau_j), D[r] = L_r(1)   (optimizer.py:855-888)."""
    tau = np.array([0.0] + collocation_points(deg, kind))
    C = np.zeros((deg + 1, deg + 1))
    D = np.zeros(deg + 1)
    for j in range(deg + 1):
        coeffs = np.poly1d([1.0])
        for r in range(deg + 1):
            if r != j:
                coeffs = coeffs * np.poly1d([1.0, -tau[r]]) / (tau[j] - tau[r])
        D[j] = coeffs(1.0)
        dp = coeffs.deriv()
        for r in range(deg + 1):
            C[j, r] = dp(tau[r])
    return tau, C, D


# ----------------------------------------------------------------------------- helpers
class _Lam:
    """Vectorised numeric callable for a list of sympy expressions."""

    def __init__(self, exprs, args):
        self.n = len(exprs)
        self.nz_idx = [i for i, e in enumerate(exprs) if e != 0]
        self.f = sp.lambdify(args, [exprs[i] for i in self.nz_idx], modules="numpy", cse=True) if self.nz_idx else None

    def __call__(self, cols, npts):
        out = np.zeros((self.n, npts))
        if self.f is not None:
            vals = self.f(*cols)
            for i, v in zip(self.nz_idx, vals):
                out[i] = v
        return out


class OracleNLP:
    def __init__(self, case):
        self.case = c = case
        self.nx, self.nu, self.np_ = len(c["x"]), len(c["u"]), len(c["p"])
        self.ntvp = len(c.get("tvp", ()))
        self.nq = self.np_ + self.ntvp            # per-edge parameter columns: scenario parameters, then the stage's _tvp
        nx, nu = self.nx, self.nu
        self.N = N = c["n_horizon"]
        self.discrete = c["model_type"] == "discrete"
        self.deg, self.ni = c["collocation_deg"], c["collocation_ni"]
        self.M = M = 0 if self.discrete else (self.deg + 1) * self.ni
        self.p_values = p_scenarios(c)
        self.n_comb = n_comb = self.p_values.shape[0]
        n_robust = c["n_robust"]
        self.S = S = n_comb ** n_robust
        self.S_u = 1 if c["open_loop"] else S
        self.nl = c["nl_cons"]
        self.ne = len(self.nl)
        self.soft = [i for i, nc in enumerate(self.nl) if nc["soft"]]
        self.n_slack = len(self.soft)
        self.n_eps = 1 if c["nl_cons_single_slack"] else N
        self.sx, self.su = np.asarray(c["x_scaling"], float), np.asarray(c["u_scaling"], float)
        self.h = c["t_step"] / self.ni
        if not self.discrete:
            self.tau, self.C, self.D = collocation_coeffs(self.deg, c["collocation_type"])

        # ---- layout
        self.off_u = (N + 1) * S * (M + 1) * nx
        self.off_eps = self.off_u + N * self.S_u * nu
        self.n_opt_x = self.off_eps + self.n_eps * S * self.n_slack
        self.p_off_x0 = 0
        self.p_off_tvp = nx                         # _tvp: N+1 stages (_mpc.py:1160-1165)
        self.p_off_p = nx + (N + 1) * self.ntvp
        self.p_off_uprev = self.p_off_p + n_comb * self.np_
        self.n_opt_p = self.p_off_uprev + nu

        # ---- tree (optimizer.py:1011-1048)
        self.n_branches = [n_comb if k < n_robust else 1 for k in range(N)]
        self.n_scen = [n_comb ** min(k, n_robust) for k in range(N + 1)]
        edges = []
        self.parent = -np.ones((N + 1, S), int)
        for k in range(N):
            cnt = 0
            for s in range(self.n_scen[k]):
                boff = 0 if (n_robust == 0 or k < n_robust) else s % self.n_branches[0]
                for b in range(self.n_branches[k]):
                    child = cnt
                    self.parent[k + 1, child] = s
                    cnt += 1
                    edges.append((k, s, b, child, b + boff, 0 if c["open_loop"] else s))
        self.edges = np.array(edges, int)
        self.E = E = len(edges)
        self.rows_per_edge = M * nx + nx + self.ne if not self.discrete else nx + self.ne
        self.n_g = nx + E * self.rows_per_edge
        self.omega = np.array([1.0 / self.n_scen[k + 1] for k in range(N)])

        self._build_functions()
        self._build_bounds()
        self._build_index()

    # ------------------------------------------------------------------ index helpers
    def ix(self, k, s, c):
        return ((k * self.S + s) * (self.M + 1) + c) * self.nx

    def iu(self, k, s):
        return self.off_u + (k * self.S_u + s) * self.nu

    def ieps(self, e, s):
        return self.off_eps + (e * self.S + s) * self.n_slack

    def slot(self, i, r):
        return r - 1 if i == 0 else self.deg + (i - 1) * (self.deg + 1) + r

    # ------------------------------------------------------------------ sympy -> numpy
    def _build_functions(self):
        c = self.case
        nx, nu = self.nx, self.nu
        xs = sp.symbols(f"xs0:{nx}")
        us = sp.symbols(f"us0:{nu}")
        ps = sp.symbols(f"pp0:{self.np_}") if self.np_ else ()
        sub = {c["x"][i]: xs[i] * float(self.sx[i]) for i in range(nx)}
        sub.update({c["u"][i]: us[i] * float(self.su[i]) for i in range(nu)})
        sub.update({c["p"][i]: ps[i] for i in range(self.np_)})
        tv = sp.symbols(f"tv0:{self.ntvp}") if self.ntvp else ()
        sub.update({c["tvp"][i]: tv[i] for i in range(self.ntvp)})
        ps = tuple(ps) + tuple(tv)
        v = list(xs) + list(us)
        args = list(xs) + list(us) + list(ps)
        scale = 1.0 if self.discrete else self.h
        F = [sp.sympify(e).subs(sub) * scale / float(self.sx[i]) for i, e in enumerate(c["rhs"])]
        self.F = _Lam(F, args)
        self.JF = _Lam([sp.diff(f, a) for f in F for a in v], args)
        self.HF = _Lam([sp.diff(f, a, b) for f in F for a in v for b in v], args)
        L = sp.sympify(c["lterm"]).subs(sub)
        self.L = _Lam([L], args)
        self.gL = _Lam([sp.diff(L, a) for a in v], args)
        self.HL = _Lam([sp.diff(L, a, b) for a in v for b in v], args)
        Mt = sp.sympify(c["mterm"]).subs(sub)
        argm = list(xs) + list(ps)
        self.Mt = _Lam([Mt], argm)
        self.gM = _Lam([sp.diff(Mt, a) for a in xs], argm)
        self.HM = _Lam([sp.diff(Mt, a, b) for a in xs for b in xs], argm)
        G = [sp.sympify(nc["expr"]).subs(sub) for nc in self.nl]
        self.G = _Lam(G, args)
        self.JG = _Lam([sp.diff(g, a) for g in G for a in v], args)
        self.HG = _Lam([sp.diff(g, a, b) for g in G for a in v for b in v], args)
        # user-defined input penalty rterm(x, u, u_prev, tvp, p) (_mpc.py:593-677): evaluated with unscaled x, u and the SCALED
        # previous input (_mpc.py:1263-1269); case key "rterm_expr" over the symbols case["u_prev"]
        self.R = None
        if c.get("rterm_expr") is not None:
            ups = sp.symbols(f"ups0:{nu}")
            subr = dict(sub)
            subr.update({c["u_prev"][i]: ups[i] for i in range(nu)})
            Rx = sp.sympify(c["rterm_expr"]).subs(subr)
            vr = list(xs) + list(us) + list(ups)
            argr = vr + list(ps)
            self.R = _Lam([Rx], argr)
            self.gR = _Lam([sp.diff(Rx, a) for a in vr], argr)
            self.HR = _Lam([sp.diff(Rx, a, b) for a in vr for b in vr], argr)
        self.aux = {k: sp.lambdify(list(c["x"]) + list(c["u"]) + list(c["p"]) + list(c.get("tvp", ())), e, "numpy")
                    for k, e in c["aux"].items()}

    # ------------------------------------------------------------------ bounds (_mpc.py:1061-1095)
    def _build_bounds(self):
        c = self.case
        N, S, M, nx, nu = self.N, self.S, self.M, self.nx, self.nu
        lb = -np.inf * np.ones(self.n_opt_x)
        ub = np.inf * np.ones(self.n_opt_x)
        xl, xu = np.asarray(c["x_lb"], float) / self.sx, np.asarray(c["x_ub"], float) / self.sx
        X_lb = lb[:self.off_u].reshape(N + 1, S, M + 1, nx)
        X_ub = ub[:self.off_u].reshape(N + 1, S, M + 1, nx)
        if c["cons_check_colloc_points"]:
            X_lb[1:N] = xl
            X_ub[1:N] = xu
        else:
            X_lb[1:N, :, -1] = xl
            X_ub[1:N, :, -1] = xu
        if c["use_terminal_bounds"]:
            X_lb[N, :, -1] = xl
            X_ub[N, :, -1] = xu
        else:
            X_lb[N, :, -1] = -np.inf
            X_ub[N, :, -1] = np.inf
        lb[self.off_u:self.off_eps].reshape(-1, nu)[:] = np.asarray(c["u_lb"], float) / self.su
        ub[self.off_u:self.off_eps].reshape(-1, nu)[:] = np.asarray(c["u_ub"], float) / self.su
        if self.n_slack:
            lb[self.off_eps:].reshape(-1, self.n_slack)[:] = 0.0
            ub[self.off_eps:].reshape(-1, self.n_slack)[:] = [self.nl[i]["max_violation"] for i in self.soft]
        self.lbx, self.ubx = lb, ub
        lbg = np.zeros(self.n_g)
        ubg = np.zeros(self.n_g)
        if self.ne:
            r0 = self.nx + (self.rows_per_edge - self.ne)
            for e in range(self.E):
                sl = slice(r0 + e * self.rows_per_edge, r0 + e * self.rows_per_edge + self.ne)
                lbg[sl] = -np.inf
                ubg[sl] = [nc["ub"] for nc in self.nl]
        self.lbg, self.ubg = lbg, ubg

    # ------------------------------------------------------------------ static index arrays
    def _build_index(self):
        nx, nu, M, deg, ni = self.nx, self.nu, self.M, self.deg, self.ni
        E = self.E
        k, s, b, ch, pidx, su = self.edges.T
        self.col_xpar = np.array([self.ix(kk, ss, M) for kk, ss in zip(k, s)])        # parent node state
        self.col_u = np.array([self.iu(kk, ss) for kk, ss in zip(k, su)])
        self.col_xch = np.array([self.ix(kk + 1, cc, M) for kk, cc in zip(k, ch)])    # child node state
        self.col_blk = np.array([self.ix(kk + 1, cc, 0) for kk, cc in zip(k, ch)])    # child slot 0
        self.row0 = nx + np.arange(E) * self.rows_per_edge
        keps = np.minimum(k, self.n_eps - 1)
        self.col_eps = np.array([self.ieps(ke, ss) for ke, ss in zip(keps, s)]) if self.n_slack else None
        # previous-input column for rterm: (k>0) u[k-1, parent[k][s_u]]
        self.col_uprev = np.array([self.iu(kk - 1, self.parent[kk, ss]) if kk > 0 else -1 for kk, ss in zip(k, su)])
        self.pidx = pidx
        if not self.discrete:
            # column of point (i,r) for each edge: shape (E, ni, deg+1)
            cp = np.zeros((E, ni, deg + 1), int)
            for i in range(ni):
                for r in range(deg + 1):
                    if i == 0 and r == 0:
                        cp[:, i, r] = self.col_xpar
                    else:
                        cp[:, i, r] = self.col_blk + self.slot(i, r) * nx
            self.col_pt = cp
            nxt = np.zeros((E, ni), int)
            for i in range(ni):
                nxt[:, i] = self.col_blk + (self.slot(i + 1, 0) if i + 1 < ni else M - 1) * nx
            self.col_next = nxt

    # ------------------------------------------------------------------ evaluation pieces
    def _pvals(self, p, terminal=False):
        """(E, nq): the edge's scenario parameters and the _tvp of its stage k (of stage k+1 for the terminal cost:
        _mpc.py:1254-1256 evaluates mterm with opt_p['_tvp', n_horizon])."""
        P = p[self.p_off_p:self.p_off_uprev].reshape(self.n_comb, self.np_)[self.pidx]
        if self.ntvp:
            T = p[self.p_off_tvp:self.p_off_p].reshape(self.N + 1, self.ntvp)
            P = np.concatenate([P, T[self.edges[:, 0] + (1 if terminal else 0)]], axis=1)
        return P

    def _stage_cols(self, x, p):
        """columns for (x_parent,u,p) per edge -> list of 1-D arrays."""
        nx, nu = self.nx, self.nu
        Xp = x[self.col_xpar[:, None] + np.arange(nx)]
        U = x[self.col_u[:, None] + np.arange(nu)]
        P = self._pvals(p)
        return Xp, U, P

    def _coll_cols(self, x, p):
        nx, nu = self.nx, self.nu
        Xpt = x[self.col_pt[..., None] + np.arange(nx)]            # (E,ni,deg+1,nx)
        U = x[self.col_u[:, None] + np.arange(nu)]                  # (E,nu)
        P = self._pvals(p)
        return Xpt, U, P

    def _pt_args(self, Xpt, U, P):
        """flatten collocation points (E,ni,j=1..deg) into arg columns"""
        E, ni, deg = self.E, self.ni, self.deg
        xs = Xpt[:, :, 1:, :].reshape(E * ni * deg, self.nx)
        us = np.repeat(U, ni * deg, axis=0)
        ps = np.repeat(P, ni * deg, axis=0)
        cols = [xs[:, i] for i in range(self.nx)] + [us[:, i] for i in range(self.nu)] + [ps[:, i] for i in range(self.nq)]
        return cols, E * ni * deg

    def _st_args(self, Xp, U, P):
        return [Xp[:, i] for i in range(self.nx)] + [U[:, i] for i in range(self.nu)] + [P[:, i] for i in range(self.nq)], self.E

    # ------------------------------------------------------------------ NLP functions
    def f(self, x, p):
        Xp, U, P = self._stage_cols(x, p)
        cols, n = self._st_args(Xp, U, P)
        k = self.edges[:, 0]
        w = self.omega[k]
        obj = np.sum(w * self.L(cols, n)[0])
        last = k == self.N - 1
        Xc = x[self.col_xch[:, None] + np.arange(self.nx)]
        Pm = self._pvals(p, terminal=True)
        colm = [Xc[:, i] for i in range(self.nx)] + [Pm[:, i] for i in range(self.nq)]
        obj += np.sum((w * self.Mt(colm, n)[0])[last])
        up = p[self.p_off_uprev:] / self.su
        Uprev = np.where((self.col_uprev >= 0)[:, None], x[np.maximum(self.col_uprev, 0)[:, None] + np.arange(self.nu)], up)
        if self.R is not None:
            obj += np.sum(w * self.R(self._r_args(Xp, U, Uprev, P), n)[0])
        else:
            obj += np.sum(w[:, None] * np.asarray(self.case["rterm"]) * (U - Uprev) ** 2)
        if self.n_slack:
            Eps = x[self.col_eps[:, None] + np.arange(self.n_slack)]
            pen = np.array([self.nl[i]["penalty"] for i in self.soft])
            obj += np.sum(Eps * pen)
        return float(obj)

    def _r_args(self, Xp, U, Uprev, P):
        return [Xp[:, i] for i in range(self.nx)] + [U[:, i] for i in range(self.nu)] + [Uprev[:, i] for i in range(self.nu)] + \
               [P[:, i] for i in range(self.nq)]

    def grad(self, x, p):
        nx, nu = self.nx, self.nu
        g = np.zeros(self.n_opt_x)
        Xp, U, P = self._stage_cols(x, p)
        cols, n = self._st_args(Xp, U, P)
        k = self.edges[:, 0]
        w = self.omega[k]
        gl = self.gL(cols, n) * w
        np.add.at(g, self.col_xpar[:, None] + np.arange(nx), gl[:nx].T)
        np.add.at(g, self.col_u[:, None] + np.arange(nu), gl[nx:].T)
        last = k == self.N - 1
        Xc = x[self.col_xch[:, None] + np.arange(nx)]
        Pm = self._pvals(p, terminal=True)
        colm = [Xc[:, i] for i in range(nx)] + [Pm[:, i] for i in range(self.nq)]
        gm = self.gM(colm, n) * w
        np.add.at(g, self.col_xch[last][:, None] + np.arange(nx), gm[:, last].T)
        up = p[self.p_off_uprev:] / self.su
        has = self.col_uprev >= 0
        Uprev = np.where(has[:, None], x[np.maximum(self.col_uprev, 0)[:, None] + np.arange(nu)], up)
        if self.R is not None:
            gr = (self.gR(self._r_args(Xp, U, Uprev, P), n) * w).T                   # (E, nx + 2 nu)
            np.add.at(g, self.col_xpar[:, None] + np.arange(nx), gr[:, :nx])
            np.add.at(g, self.col_u[:, None] + np.arange(nu), gr[:, nx:nx + nu])
            np.add.at(g, self.col_uprev[has][:, None] + np.arange(nu), gr[has][:, nx + nu:])
        else:
            d = 2.0 * w[:, None] * np.asarray(self.case["rterm"]) * (U - Uprev)
            np.add.at(g, self.col_u[:, None] + np.arange(nu), d)
            np.add.at(g, self.col_uprev[has][:, None] + np.arange(nu), -d[has])
        if self.n_slack:
            pen = np.array([self.nl[i]["penalty"] for i in self.soft])
            np.add.at(g, self.col_eps[:, None] + np.arange(self.n_slack), np.tile(pen, (self.E, 1)))
        return g

    def g(self, x, p):
        nx, nu, M, deg, ni = self.nx, self.nu, self.M, self.deg, self.ni
        out = np.zeros(self.n_g)
        out[:nx] = x[self.ix(0, 0, M):self.ix(0, 0, M) + nx] - p[:nx] / self.sx
        E = self.E
        G = out[nx:].reshape(E, self.rows_per_edge)
        Xp, U, P = self._stage_cols(x, p)
        Xc = x[self.col_xch[:, None] + np.arange(nx)]
        if self.discrete:
            cols, n = self._st_args(Xp, U, P)
            G[:, :nx] = self.F(cols, n).T - Xc
        else:
            Xpt, U2, P2 = self._coll_cols(x, p)
            cols, n = self._pt_args(Xpt, U2, P2)
            Fv = self.F(cols, n).T.reshape(E, ni, deg, nx)
            for i in range(ni):
                base = i * (deg + 1) * nx
                for j in range(1, deg + 1):
                    xp = np.einsum("r,erx->ex", self.C[:, j], Xpt[:, i])
                    G[:, base + (j - 1) * nx: base + j * nx] = Fv[:, i, j - 1] - xp
                xf = np.einsum("r,erx->ex", self.D, Xpt[:, i])
                Xn = x[self.col_next[:, i][:, None] + np.arange(nx)]
                G[:, base + deg * nx: base + (deg + 1) * nx] = Xn - xf
            Xkf = x[(self.col_blk + (M - 1) * nx)[:, None] + np.arange(nx)]
            G[:, M * nx:M * nx + nx] = Xkf - Xc
        if self.ne:
            cols, n = self._st_args(Xp, U, P)
            gv = self.G(cols, n).T
            if self.n_slack:
                Eps = x[self.col_eps[:, None] + np.arange(self.n_slack)]
                for q, i in enumerate(self.soft):
                    gv[:, i] -= Eps[:, q]
            G[:, self.rows_per_edge - self.ne:] = gv
        return out

    def jac(self, x, p):
        nx, nu, M, deg, ni = self.nx, self.nu, self.M, self.deg, self.ni
        nv = nx + nu
        E = self.E
        R, Cc, V = [], [], []

        def put(rows, cols, vals):
            R.append(np.asarray(rows).ravel())
            Cc.append(np.asarray(cols).ravel())
            V.append(np.asarray(vals, float).ravel())

        ar = np.arange(nx)
        put(ar, self.ix(0, 0, M) + ar, np.ones(nx))
        Xp, U, P = self._stage_cols(x, p)
        eye = np.eye(nx)
        if self.discrete:
            cols, n = self._st_args(Xp, U, P)
            J = self.JF(cols, n).T.reshape(E, nx, nv)
            rows = self.row0[:, None, None] + ar[None, :, None] + np.zeros((1, 1, nx), int)
            put(rows, self.col_xpar[:, None, None] + ar[None, None, :] + np.zeros((1, nx, 1), int), J[:, :, :nx])
            rows_u = self.row0[:, None, None] + ar[None, :, None] + np.zeros((1, 1, nu), int)
            put(rows_u, self.col_u[:, None, None] + np.arange(nu)[None, None, :] + np.zeros((1, nx, 1), int), J[:, :, nx:])
            put(self.row0[:, None] + ar, self.col_xch[:, None] + ar, -np.ones((E, nx)))
            nl_row0 = self.row0 + nx
        else:
            Xpt, U2, P2 = self._coll_cols(x, p)
            cols, n = self._pt_args(Xpt, U2, P2)
            J = self.JF(cols, n).T.reshape(E, ni, deg, nx, nv)
            for i in range(ni):
                base = self.row0 + i * (deg + 1) * nx
                for j in range(1, deg + 1):
                    r0 = base + (j - 1) * nx
                    rows = r0[:, None, None] + ar[None, :, None] + np.zeros((1, 1, nx), int)
                    put(rows, self.col_pt[:, i, j][:, None, None] + ar[None, None, :] + np.zeros((1, nx, 1), int), J[:, i, j - 1, :, :nx])
                    rows_u = r0[:, None, None] + ar[None, :, None] + np.zeros((1, 1, nu), int)
                    put(rows_u, self.col_u[:, None, None] + np.arange(nu)[None, None, :] + np.zeros((1, nx, 1), int), J[:, i, j - 1, :, nx:])
                    for r in range(deg + 1):
                        put(r0[:, None] + ar, self.col_pt[:, i, r][:, None] + ar, -self.C[r, j] * np.ones((E, nx)))
                r0 = base + deg * nx
                put(r0[:, None] + ar, self.col_next[:, i][:, None] + ar, np.ones((E, nx)))
                for r in range(deg + 1):
                    put(r0[:, None] + ar, self.col_pt[:, i, r][:, None] + ar, -self.D[r] * np.ones((E, nx)))
            r0 = self.row0 + M * nx
            put(r0[:, None] + ar, (self.col_blk + (M - 1) * nx)[:, None] + ar, np.ones((E, nx)))
            put(r0[:, None] + ar, self.col_xch[:, None] + ar, -np.ones((E, nx)))
            nl_row0 = self.row0 + M * nx + nx
        if self.ne:
            cols, n = self._st_args(Xp, U, P)
            JG = self.JG(cols, n).T.reshape(E, self.ne, nv)
            an = np.arange(self.ne)
            rows = nl_row0[:, None, None] + an[None, :, None] + np.zeros((1, 1, nx), int)
            put(rows, self.col_xpar[:, None, None] + ar[None, None, :] + np.zeros((1, self.ne, 1), int), JG[:, :, :nx])
            rows = nl_row0[:, None, None] + an[None, :, None] + np.zeros((1, 1, nu), int)
            put(rows, self.col_u[:, None, None] + np.arange(nu)[None, None, :] + np.zeros((1, self.ne, 1), int), JG[:, :, nx:])
            for q, i in enumerate(self.soft):
                put(nl_row0 + i, self.col_eps + q, -np.ones(E))
        R, Cc, V = np.concatenate(R), np.concatenate(Cc), np.concatenate(V)
        keep = V != 0.0
        return sps.csr_matrix((V[keep], (R[keep], Cc[keep])), shape=(self.n_g, self.n_opt_x))

    def hess(self, x, p, sigma, lam):
        nx, nu, M, deg, ni = self.nx, self.nu, self.M, self.deg, self.ni
        nv = nx + nu
        E = self.E
        R, Cc, V = [], [], []

        def put_block(colsA, colsB, vals):
            # vals (E, nA, nB); colsA (E,nA); colsB (E,nB)
            R.append(np.broadcast_to(colsA[:, :, None], vals.shape).ravel())
            Cc.append(np.broadcast_to(colsB[:, None, :], vals.shape).ravel())
            V.append(np.asarray(vals, float).ravel())

        Xp, U, P = self._stage_cols(x, p)
        k = self.edges[:, 0]
        w = self.omega[k] * sigma
        cols, n = self._st_args(Xp, U, P)
        vcols = np.concatenate([self.col_xpar[:, None] + np.arange(nx), self.col_u[:, None] + np.arange(nu)], axis=1)
        HL = self.HL(cols, n).T.reshape(E, nv, nv) * w[:, None, None]
        put_block(vcols, vcols, HL)
        last = k == self.N - 1
        Xc = x[self.col_xch[:, None] + np.arange(nx)]
        Pm = self._pvals(p, terminal=True)
        colm = [Xc[:, i] for i in range(nx)] + [Pm[:, i] for i in range(self.nq)]
        HM = self.HM(colm, n).T.reshape(E, nx, nx) * w[:, None, None]
        cc = self.col_xch[:, None] + np.arange(nx)
        put_block(cc[last], cc[last], HM[last])
        # rterm
        ucols = self.col_u[:, None] + np.arange(nu)
        has = self.col_uprev >= 0
        pcols = self.col_uprev[has][:, None] + np.arange(nu)
        if self.R is not None:
            U = x[ucols]
            up = p[self.p_off_uprev:] / self.su
            Uprev = np.where(has[:, None], x[np.maximum(self.col_uprev, 0)[:, None] + np.arange(nu)], up)
            HR = self.HR(self._r_args(Xp, U, Uprev, P), n).T.reshape(E, nx + 2 * nu, nx + 2 * nu) * w[:, None, None]
            xu = np.concatenate([self.col_xpar[:, None] + np.arange(nx), ucols], axis=1)       # (x, u) part on every edge
            put_block(xu, xu, HR[:, :nx + nu, :nx + nu])
            put_block(xu[has], pcols, HR[has][:, :nx + nu, nx + nu:])                         # u_prev is a variable for k > 0
            put_block(pcols, xu[has], HR[has][:, nx + nu:, :nx + nu])
            put_block(pcols, pcols, HR[has][:, nx + nu:, nx + nu:])
        else:
            r2 = 2.0 * w[:, None] * np.asarray(self.case["rterm"])
            R.append(ucols.ravel()); Cc.append(ucols.ravel()); V.append(r2.ravel())
            R.append(pcols.ravel()); Cc.append(pcols.ravel()); V.append(r2[has].ravel())
            R.append(ucols[has].ravel()); Cc.append(pcols.ravel()); V.append(-r2[has].ravel())
            R.append(pcols.ravel()); Cc.append(ucols[has].ravel()); V.append(-r2[has].ravel())
        lamE = lam[nx:].reshape(E, self.rows_per_edge)
        if self.discrete:
            HF = self.HF(cols, n).T.reshape(E, nx, nv, nv)
            H = np.einsum("ei,eiab->eab", lamE[:, :nx], HF)
            put_block(vcols, vcols, H)
        else:
            Xpt, U2, P2 = self._coll_cols(x, p)
            colsp, npt = self._pt_args(Xpt, U2, P2)
            HF = self.HF(colsp, npt).T.reshape(E, ni, deg, nx, nv, nv)
            for i in range(ni):
                for j in range(1, deg + 1):
                    r0 = i * (deg + 1) * nx + (j - 1) * nx
                    H = np.einsum("ei,eiab->eab", lamE[:, r0:r0 + nx], HF[:, i, j - 1])
                    vc = np.concatenate([self.col_pt[:, i, j][:, None] + np.arange(nx), self.col_u[:, None] + np.arange(nu)], axis=1)
                    put_block(vc, vc, H)
        if self.ne:
            HG = self.HG(cols, n).T.reshape(E, self.ne, nv, nv)
            H = np.einsum("ei,eiab->eab", lamE[:, self.rows_per_edge - self.ne:], HG)
            put_block(vcols, vcols, H)
        R, Cc, V = np.concatenate(R), np.concatenate(Cc), np.concatenate(V)
        keep = V != 0.0
        return sps.csr_matrix((V[keep], (R[keep], Cc[keep])), shape=(self.n_opt_x, self.n_opt_x))

    # ------------------------------------------------------------------ protocol helpers
    def opt_p(self, x0, u_prev=None, tvp=None):
        p = np.zeros(self.n_opt_p)
        p[:self.nx] = np.asarray(x0, float).ravel()
        if tvp is not None:
            p[self.p_off_tvp:self.p_off_p] = np.asarray(tvp, float).reshape(self.N + 1, self.ntvp).ravel()
        p[self.p_off_p:self.p_off_uprev] = self.p_values.ravel()
        if u_prev is not None:
            p[self.p_off_uprev:] = np.asarray(u_prev, float).ravel()
        return p

    def initial_guess(self, x0, u0=None):
        """MPC.set_initial_guess (_mpc.py:969-971): every _x slot = x0/x_scaling, every _u = u0/u_scaling."""
        x = np.zeros(self.n_opt_x)
        x[:self.off_u].reshape(-1, self.nx)[:] = np.asarray(x0, float).ravel() / self.sx
        if u0 is not None:
            x[self.off_u:self.off_eps].reshape(-1, self.nu)[:] = np.asarray(u0, float).ravel() / self.su
        return x

    def scaling_vector(self):
        s = np.ones(self.n_opt_x)
        s[:self.off_u].reshape(-1, self.nx)[:] = self.sx
        s[self.off_u:self.off_eps].reshape(-1, self.nu)[:] = self.su
        return s

    def u0_of(self, x):
        return x[self.iu(0, 0):self.iu(0, 0) + self.nu] * self.su
