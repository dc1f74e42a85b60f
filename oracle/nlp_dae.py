"""Flat sparse restatement of the reference's NLP for models WITH algebraic states (`_z`, DAE) - TEST INFRASTRUCTURE ONLY.

Same contract as oracle/nlp.py:OracleNLP (the class the oracle's interior-point method and the parity tests talk to), written
for the general interval function of the reference instead of the ODE special case:
  variables        /root/reference/do_mpc/controller/_mpc.py:1126-1134   opt_x = [_x | _z | _u | _eps]
  interval rows    /root/reference/do_mpc/optimizer.py:905-983           per finite element: alg(point 0), then per collocation
                                                                         point [collocation rows, alg rows], then continuity;
                                                                         discrete models [alg ; rhs]  (:820-824)
  use of z         _mpc.py:1213 (dynamics: `_z[k, child, :]`), :1241 (nl_cons: `_z[k, s, 0]`), :1252 (lterm: `_z[k, s, -1]`)
  scaling          optimizer.py:804-818 (x, u, z scaled; rhs / x_scaling; alg rows unscaled)
Point functions (rhs+alg, lterm, mterm, nl_cons) and their first / second derivatives come from sympy; the global functions
are assembled by scatter, vectorised over the edges with plain loops over finite elements and points.  No structure is
exploited by the solver that uses this class (oracle/ipm.py: general sparse LU of the KKT matrix).
"""
import numpy as np
import scipy.sparse as sps
import sympy as sp

from .models import p_scenarios
from .nlp import _Lam, collocation_coeffs


class OracleNLPDae:
    def __init__(self, case):
        self.case = c = dict(case)
        c.setdefault("z", ()); c.setdefault("alg", ())        # (a model without algebraic states: nl_cons rows at the collocation points)
        self.nx, self.nu, self.nz, self.np_ = len(c["x"]), len(c["u"]), len(c["z"]), len(c["p"])
        self.ntvp = len(c.get("tvp", ()))
        self.nq = self.np_ + self.ntvp
        nx, nu, nz = self.nx, self.nu, self.nz
        self.N = N = c["n_horizon"]
        self.discrete = c["model_type"] == "discrete"
        self.deg, self.ni = c["collocation_deg"], c["collocation_ni"]
        self.M = M = 0 if self.discrete else (self.deg + 1) * self.ni
        self.MZ = max(M, 1)
        self.p_values = p_scenarios(c)
        self.n_comb = n_comb = self.p_values.shape[0]
        n_robust = c["n_robust"]
        self.S = S = n_comb ** n_robust
        assert not c["open_loop"]
        self.nl = c["nl_cons"]
        # nl_cons rows per edge: one evaluation at (x_n, u, z first point), or with nl_cons_check_colloc_points one per stored
        # point i of the interval at (`_x[k+1, s, i]`, u, `_z[k, s, i]`), s = the PARENT's scenario index (_mpc.py:1229-1246)
        self.neb = len(self.nl)
        self.nlb = M if (c["nl_cons_check_colloc_points"] and self.neb and not self.discrete) else 1
        self.nl_colloc = self.nlb > 1 or bool(c["nl_cons_check_colloc_points"] and self.neb and not self.discrete)
        self.ne = self.neb * self.nlb
        self.soft = [i for i, nc in enumerate(self.nl) if nc["soft"]]
        self.n_slack = len(self.soft)
        self.n_eps = 1 if c["nl_cons_single_slack"] else N       # (_mpc.py:1120-1123)
        self.sx, self.su = np.asarray(c["x_scaling"], float), np.asarray(c["u_scaling"], float)
        self.sz = np.asarray(c.get("z_scaling", np.ones(nz)), float)
        self.h = c["t_step"] / self.ni
        if not self.discrete:
            self.tau, self.C, self.D = collocation_coeffs(self.deg, c["collocation_type"])
        # ---- layout (_mpc.py:1126-1134)
        self.off_z = (N + 1) * S * (M + 1) * nx
        self.off_u = self.off_z + N * S * self.MZ * nz
        self.off_eps = self.off_u + N * S * nu
        self.n_opt_x = self.off_eps + self.n_eps * S * self.n_slack
        self.p_off_tvp = nx
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
                    edges.append((k, s, b, child, b + boff))
        self.edges = np.array(edges, int)
        self.E = E = len(edges)
        self.ELR = nz + self.deg * (nx + nz) + nx                       # rows of one finite element
        self.rows_block = (self.ni * self.ELR) if not self.discrete else nz
        self.rows_per_edge = self.rows_block + nx + self.ne
        self.n_g = nx + E * self.rows_per_edge
        self.omega = np.array([1.0 / self.n_scen[k + 1] for k in range(N)])
        self._build_functions()
        self._build_bounds()
        self._build_index()

    # ------------------------------------------------------------------ index helpers
    def ix(self, k, s, c):
        return ((k * self.S + s) * (self.M + 1) + c) * self.nx

    def iz(self, k, s, c):
        return self.off_z + ((k * self.S + s) * self.MZ + c) * self.nz

    def iu(self, k, s):
        return self.off_u + (k * self.S + s) * self.nu

    def ieps(self, e, s):
        return self.off_eps + (e * self.S + s) * self.n_slack

    def slot(self, i, r):
        return r - 1 if i == 0 else self.deg + (i - 1) * (self.deg + 1) + r

    # ------------------------------------------------------------------ sympy -> numpy
    def _build_functions(self):
        c = self.case
        nx, nu, nz = self.nx, self.nu, self.nz
        xs, us, zs = sp.symbols(f"xs0:{nx}"), sp.symbols(f"us0:{nu}"), sp.symbols(f"zs0:{nz}")
        ps = sp.symbols(f"pp0:{self.np_}") if self.np_ else ()
        tv = sp.symbols(f"tv0:{self.ntvp}") if self.ntvp else ()
        sub = {c["x"][i]: xs[i] * float(self.sx[i]) for i in range(nx)}
        sub.update({c["u"][i]: us[i] * float(self.su[i]) for i in range(nu)})
        sub.update({c["z"][i]: zs[i] * float(self.sz[i]) for i in range(nz)})
        sub.update({c["p"][i]: ps[i] for i in range(self.np_)})
        sub.update({c["tvp"][i]: tv[i] for i in range(self.ntvp)})
        pq = tuple(ps) + tuple(tv)
        v = list(xs) + list(us) + list(zs)
        args = v + list(pq)
        scale = 1.0 if self.discrete else self.h
        F = [sp.sympify(e).subs(sub) * scale / float(self.sx[i]) for i, e in enumerate(c["rhs"])]
        F += [sp.sympify(e).subs(sub) for e in c["alg"]]
        self.F = _Lam(F, args)
        self.JF = _Lam([sp.diff(f, a) for f in F for a in v], args)
        self.HF = _Lam([sp.diff(f, a, b) for f in F for a in v for b in v], args)
        L = sp.sympify(c["lterm"]).subs(sub)
        self.L, self.gL = _Lam([L], args), _Lam([sp.diff(L, a) for a in v], args)
        self.HL = _Lam([sp.diff(L, a, b) for a in v for b in v], args)
        Mt = sp.sympify(c["mterm"]).subs(sub)
        argm = list(xs) + list(pq)
        self.Mt, self.gM = _Lam([Mt], argm), _Lam([sp.diff(Mt, a) for a in xs], argm)
        self.HM = _Lam([sp.diff(Mt, a, b) for a in xs for b in xs], argm)
        G = [sp.sympify(nc["expr"]).subs(sub) for nc in self.nl]
        self.G, self.JG = _Lam(G, args), _Lam([sp.diff(g, a) for g in G for a in v], args)
        self.HG = _Lam([sp.diff(g, a, b) for g in G for a in v for b in v], args)
        self.nav, self.nf = nx + nu + nz, nx + nz

    # ------------------------------------------------------------------ bounds (_mpc.py:1061-1095)
    def _build_bounds(self):
        c = self.case
        N, S, M, nx, nu = self.N, self.S, self.M, self.nx, self.nu
        lb, ub = -np.inf * np.ones(self.n_opt_x), np.inf * np.ones(self.n_opt_x)
        xl, xu = np.asarray(c["x_lb"], float) / self.sx, np.asarray(c["x_ub"], float) / self.sx
        X_lb, X_ub = lb[:self.off_z].reshape(N + 1, S, M + 1, nx), ub[:self.off_z].reshape(N + 1, S, M + 1, nx)
        if c["cons_check_colloc_points"]:
            X_lb[1:N], X_ub[1:N] = xl, xu
        else:
            X_lb[1:N, :, -1], X_ub[1:N, :, -1] = xl, xu
        if c["use_terminal_bounds"]:
            X_lb[N, :, -1], X_ub[N, :, -1] = xl, xu
        if self.nz and "z_lb" in c:
            zl, zu = np.asarray(c["z_lb"], float) / self.sz, np.asarray(c["z_ub"], float) / self.sz
            Z_lb, Z_ub = lb[self.off_z:self.off_u].reshape(N, S, self.MZ, self.nz), ub[self.off_z:self.off_u].reshape(N, S, self.MZ, self.nz)
            if c["cons_check_colloc_points"]:
                Z_lb[:], Z_ub[:] = zl, zu
            else:
                Z_lb[:, :, 0], Z_ub[:, :, 0] = zl, zu
        lb[self.off_u:self.off_eps].reshape(-1, nu)[:] = np.asarray(c["u_lb"], float) / self.su
        ub[self.off_u:self.off_eps].reshape(-1, nu)[:] = np.asarray(c["u_ub"], float) / self.su
        if self.n_slack:
            lb[self.off_eps:].reshape(-1, self.n_slack)[:] = 0.0
            ub[self.off_eps:].reshape(-1, self.n_slack)[:] = [self.nl[i]["max_violation"] for i in self.soft]
        self.lbx, self.ubx = lb, ub
        lbg, ubg = np.zeros(self.n_g), np.zeros(self.n_g)
        if self.ne:
            r0 = self.nx + (self.rows_per_edge - self.ne)
            for e in range(self.E):
                sl = slice(r0 + e * self.rows_per_edge, r0 + e * self.rows_per_edge + self.ne)
                lbg[sl] = -np.inf
                ubg[sl] = np.tile([nc["ub"] for nc in self.nl], self.nlb)
        self.lbg, self.ubg = lbg, ubg

    # ------------------------------------------------------------------ static index arrays (one row per edge)
    def _build_index(self):
        nx, nu, nz, M, deg, ni = self.nx, self.nu, self.nz, self.M, self.deg, self.ni
        k, s, b, ch, pidx = self.edges.T
        self.pidx = pidx
        self.col_xpar = np.array([self.ix(kk, ss, M) for kk, ss in zip(k, s)])
        self.col_u = np.array([self.iu(kk, ss) for kk, ss in zip(k, s)])
        self.col_xch = np.array([self.ix(kk + 1, cc, M) for kk, cc in zip(k, ch)])
        self.col_blk = np.array([self.ix(kk + 1, cc, 0) for kk, cc in zip(k, ch)])
        self.col_zdyn = np.array([self.iz(kk, cc, 0) for kk, cc in zip(k, ch)])          # `_z[k, child, :]`
        self.col_zl = np.array([self.iz(kk, ss, self.MZ - 1) for kk, ss in zip(k, s)])    # `_z[k, s, -1]` (stage cost)
        self.col_zn = np.array([self.iz(kk, ss, 0) for kk, ss in zip(k, s)])              # `_z[k, s, 0]` (nl_cons)
        self.row0 = nx + np.arange(self.E) * self.rows_per_edge
        self.col_eps = np.array([self.ieps(min(kk, self.n_eps - 1), ss) for kk, ss in zip(k, s)]) if self.n_slack else None     # (_mpc.py:1228)
        self.col_uprev = np.array([self.iu(kk - 1, self.parent[kk, ss]) if kk > 0 else -1 for kk, ss in zip(k, s)])

    def _nl_cols(self, blk):
        """opt_x columns of the (x, u, z) inputs of nl_cons evaluation `blk` of every edge"""
        if not self.nl_colloc:
            return self.col_xpar, self.col_u, self.col_zn
        k, s = self.edges[:, 0], self.edges[:, 1]
        return (np.array([self.ix(kk + 1, ss, blk) for kk, ss in zip(k, s)]), self.col_u,
                np.array([self.iz(kk, ss, blk) for kk, ss in zip(k, s)]))

    def _col_x(self, el, j):
        return self.col_xpar if (self.discrete or (el == 0 and j == 0)) else self.col_blk + self.slot(el, j) * self.nx

    def _col_z(self, el, j):
        return self.col_zdyn + (0 if self.discrete else (el * (self.deg + 1) + j)) * self.nz

    def _col_next(self, el):
        return self.col_blk + (self.slot(el + 1, 0) if el + 1 < self.ni else self.M - 1) * self.nx

    def _rows(self, el, j):
        """(collocation rows or None, algebraic rows) of point (el, j), relative to the edge's first row"""
        nx, nz = self.nx, self.nz
        if self.discrete:
            return None, np.arange(nz)
        base = el * self.ELR
        if j == 0:
            return None, base + np.arange(nz)
        r = base + nz + (j - 1) * (nx + nz)
        return r + np.arange(nx), r + nx + np.arange(nz)

    # ------------------------------------------------------------------ evaluation helpers
    def _pvals(self, p, terminal=False):
        P = p[self.p_off_p:self.p_off_uprev].reshape(self.n_comb, self.np_)[self.pidx]
        if self.ntvp:
            T = p[self.p_off_tvp:self.p_off_p].reshape(self.N + 1, self.ntvp)
            P = np.concatenate([P, T[self.edges[:, 0] + (1 if terminal else 0)]], axis=1)
        return P

    def _args(self, x, colx, colu, colz, P):
        cols = [x[colx + i] for i in range(self.nx)] + [x[colu + i] for i in range(self.nu)] + [x[colz + i] for i in range(self.nz)]
        return cols + [P[:, i] for i in range(self.nq)], self.E

    def _vcols(self, colx, colu, colz):
        """(E, nav) global columns of the inputs (x, u, z) of a point function"""
        return np.concatenate([colx[:, None] + np.arange(self.nx), colu[:, None] + np.arange(self.nu),
                               colz[:, None] + np.arange(self.nz)], axis=1)

    def _points(self):
        if self.discrete:
            return [(0, 0)]
        return [(el, j) for el in range(self.ni) for j in range(self.deg + 1)]

    # ------------------------------------------------------------------ NLP functions
    def f(self, x, p):
        P = self._pvals(p)
        k = self.edges[:, 0]
        w = self.omega[k]
        cols, n = self._args(x, self.col_xpar, self.col_u, self.col_zl, P)
        obj = np.sum(w * self.L(cols, n)[0])
        last = k == self.N - 1
        Pm = self._pvals(p, terminal=True)
        colm = [x[self.col_xch + i] for i in range(self.nx)] + [Pm[:, i] for i in range(self.nq)]
        obj += np.sum((w * self.Mt(colm, n)[0])[last])
        U = x[self.col_u[:, None] + np.arange(self.nu)]
        up = p[self.p_off_uprev:] / self.su
        Uprev = np.where((self.col_uprev >= 0)[:, None], x[np.maximum(self.col_uprev, 0)[:, None] + np.arange(self.nu)], up)
        obj += np.sum(w[:, None] * np.asarray(self.case["rterm"]) * (U - Uprev) ** 2)
        if self.n_slack:
            Eps = x[self.col_eps[:, None] + np.arange(self.n_slack)]
            obj += np.sum(Eps * np.array([self.nl[i]["penalty"] for i in self.soft]))
        return float(obj)

    def grad(self, x, p):
        nx, nu = self.nx, self.nu
        g = np.zeros(self.n_opt_x)
        P = self._pvals(p)
        k = self.edges[:, 0]
        w = self.omega[k]
        cols, n = self._args(x, self.col_xpar, self.col_u, self.col_zl, P)
        gl = self.gL(cols, n) * w
        np.add.at(g, self._vcols(self.col_xpar, self.col_u, self.col_zl), gl.T)
        last = k == self.N - 1
        Pm = self._pvals(p, terminal=True)
        colm = [x[self.col_xch + i] for i in range(nx)] + [Pm[:, i] for i in range(self.nq)]
        gm = self.gM(colm, n) * w
        np.add.at(g, self.col_xch[last][:, None] + np.arange(nx), gm[:, last].T)
        U = x[self.col_u[:, None] + np.arange(nu)]
        up = p[self.p_off_uprev:] / self.su
        has = self.col_uprev >= 0
        Uprev = np.where(has[:, None], x[np.maximum(self.col_uprev, 0)[:, None] + np.arange(nu)], up)
        d = 2.0 * w[:, None] * np.asarray(self.case["rterm"]) * (U - Uprev)
        np.add.at(g, self.col_u[:, None] + np.arange(nu), d)
        np.add.at(g, self.col_uprev[has][:, None] + np.arange(nu), -d[has])
        if self.n_slack:
            pen = np.array([self.nl[i]["penalty"] for i in self.soft])
            np.add.at(g, self.col_eps[:, None] + np.arange(self.n_slack), np.tile(pen, (self.E, 1)))
        return g

    def g(self, x, p):
        nx, nz, M, deg = self.nx, self.nz, self.M, self.deg
        out = np.zeros(self.n_g)
        out[:nx] = x[self.ix(0, 0, M):self.ix(0, 0, M) + nx] - p[:nx] / self.sx
        G = out[nx:].reshape(self.E, self.rows_per_edge)
        P = self._pvals(p)
        Xc = x[self.col_xch[:, None] + np.arange(nx)]
        for (el, j) in self._points():
            cols, n = self._args(x, self._col_x(el, j), self.col_u, self._col_z(el, j), P)
            Fv = self.F(cols, n).T                                        # (E, nf)
            rc, ra = self._rows(el, j)
            G[:, ra] = Fv[:, nx:]
            if self.discrete:
                G[:, nz:nz + nx] = Fv[:, :nx] - Xc
            elif rc is not None:
                xp = sum(self.C[r, j] * x[self._col_x(el, r)[:, None] + np.arange(nx)] for r in range(deg + 1))
                G[:, rc] = Fv[:, :nx] - xp
        if not self.discrete:
            for el in range(self.ni):
                xf = sum(self.D[r] * x[self._col_x(el, r)[:, None] + np.arange(nx)] for r in range(deg + 1))
                rr = el * self.ELR + nz + deg * (nx + nz) + np.arange(nx)
                G[:, rr] = x[self._col_next(el)[:, None] + np.arange(nx)] - xf
            G[:, self.rows_block:self.rows_block + nx] = x[(self.col_blk + (M - 1) * nx)[:, None] + np.arange(nx)] - Xc
        for blk in range(self.nlb if self.ne else 0):
            cols, n = self._args(x, *self._nl_cols(blk), P)
            gv = self.G(cols, n).T
            if self.n_slack:
                Eps = x[self.col_eps[:, None] + np.arange(self.n_slack)]
                for q, i in enumerate(self.soft):
                    gv[:, i] -= Eps[:, q]
            r0 = self.rows_per_edge - self.ne + blk * self.neb
            G[:, r0:r0 + self.neb] = gv
        return out

    def jac(self, x, p):
        nx, nz, M, deg, nav, nf = self.nx, self.nz, self.M, self.deg, self.nav, self.nf
        E = self.E
        R, Cc, V = [], [], []

        def put(rows, cols, vals):
            rows, cols, vals = np.broadcast_arrays(np.asarray(rows), np.asarray(cols), np.asarray(vals, float))
            R.append(rows.ravel()); Cc.append(cols.ravel()); V.append(vals.ravel())

        ar = np.arange(nx)
        put(ar, self.ix(0, 0, M) + ar, np.ones(nx))
        P = self._pvals(p)
        for (el, j) in self._points():
            colx, colz = self._col_x(el, j), self._col_z(el, j)
            cols, n = self._args(x, colx, self.col_u, colz, P)
            J = self.JF(cols, n).T.reshape(E, nf, nav)
            vc = self._vcols(colx, self.col_u, colz)                     # (E, nav)
            rc, ra = self._rows(el, j)
            put((self.row0[:, None] + ra[None, :])[:, :, None], vc[:, None, :], J[:, nx:, :])
            if self.discrete:
                put((self.row0[:, None] + nz + ar[None, :])[:, :, None], vc[:, None, :], J[:, :nx, :])
                put(self.row0[:, None] + nz + ar, self.col_xch[:, None] + ar, -np.ones((E, nx)))
            elif rc is not None:
                put((self.row0[:, None] + rc[None, :])[:, :, None], vc[:, None, :], J[:, :nx, :])
                for r in range(deg + 1):
                    put(self.row0[:, None] + rc, self._col_x(el, r)[:, None] + ar, -self.C[r, j] * np.ones((E, nx)))
        if not self.discrete:
            for el in range(self.ni):
                rr = self.row0[:, None] + el * self.ELR + nz + deg * (nx + nz) + ar
                put(rr, self._col_next(el)[:, None] + ar, np.ones((E, nx)))
                for r in range(deg + 1):
                    put(rr, self._col_x(el, r)[:, None] + ar, -self.D[r] * np.ones((E, nx)))
            rr = self.row0[:, None] + self.rows_block + ar
            put(rr, (self.col_blk + (M - 1) * nx)[:, None] + ar, np.ones((E, nx)))
            put(rr, self.col_xch[:, None] + ar, -np.ones((E, nx)))
        for blk in range(self.nlb if self.ne else 0):
            cx, cu, cz = self._nl_cols(blk)
            cols, n = self._args(x, cx, cu, cz, P)
            JG = self.JG(cols, n).T.reshape(E, self.neb, nav)
            vc = self._vcols(cx, cu, cz)
            r0 = self.row0 + self.rows_per_edge - self.ne + blk * self.neb
            put((r0[:, None] + np.arange(self.neb)[None, :])[:, :, None], vc[:, None, :], JG)
            for q, i in enumerate(self.soft):
                put(r0 + i, self.col_eps + q, -np.ones(E))
        R, Cc, V = np.concatenate(R), np.concatenate(Cc), np.concatenate(V)
        keep = V != 0.0
        return sps.csr_matrix((V[keep], (R[keep], Cc[keep])), shape=(self.n_g, self.n_opt_x))

    def hess(self, x, p, sigma, lam):
        nx, nu, nz, nav, nf = self.nx, self.nu, self.nz, self.nav, self.nf
        E = self.E
        R, Cc, V = [], [], []

        def put_block(colsA, colsB, vals):
            R.append(np.broadcast_to(colsA[:, :, None], vals.shape).ravel())
            Cc.append(np.broadcast_to(colsB[:, None, :], vals.shape).ravel())
            V.append(np.asarray(vals, float).ravel())

        P = self._pvals(p)
        k = self.edges[:, 0]
        w = self.omega[k] * sigma
        cols, n = self._args(x, self.col_xpar, self.col_u, self.col_zl, P)
        vcl = self._vcols(self.col_xpar, self.col_u, self.col_zl)
        put_block(vcl, vcl, self.HL(cols, n).T.reshape(E, nav, nav) * w[:, None, None])
        last = k == self.N - 1
        Pm = self._pvals(p, terminal=True)
        colm = [x[self.col_xch + i] for i in range(nx)] + [Pm[:, i] for i in range(self.nq)]
        HM = self.HM(colm, n).T.reshape(E, nx, nx) * w[:, None, None]
        cc = self.col_xch[:, None] + np.arange(nx)
        put_block(cc[last], cc[last], HM[last])
        r2 = 2.0 * w[:, None] * np.asarray(self.case["rterm"])
        ucols = self.col_u[:, None] + np.arange(nu)
        R.append(ucols.ravel()); Cc.append(ucols.ravel()); V.append(r2.ravel())
        has = self.col_uprev >= 0
        pcols = self.col_uprev[has][:, None] + np.arange(nu)
        R.append(pcols.ravel()); Cc.append(pcols.ravel()); V.append(r2[has].ravel())
        R.append(ucols[has].ravel()); Cc.append(pcols.ravel()); V.append(-r2[has].ravel())
        R.append(pcols.ravel()); Cc.append(ucols[has].ravel()); V.append(-r2[has].ravel())
        lamE = lam[nx:].reshape(E, self.rows_per_edge)
        for (el, j) in self._points():
            colx, colz = self._col_x(el, j), self._col_z(el, j)
            cols, n = self._args(x, colx, self.col_u, colz, P)
            HF = self.HF(cols, n).T.reshape(E, nf, nav, nav)
            rc, ra = self._rows(el, j)
            lamp = np.zeros((E, nf))
            lamp[:, nx:] = lamE[:, ra]
            if self.discrete:
                lamp[:, :nx] = lamE[:, nz:nz + nx]
            elif rc is not None:
                lamp[:, :nx] = lamE[:, rc]
            vc = self._vcols(colx, self.col_u, colz)
            put_block(vc, vc, np.einsum("ei,eiab->eab", lamp, HF))
        for blk in range(self.nlb if self.ne else 0):
            cx, cu, cz = self._nl_cols(blk)
            cols, n = self._args(x, cx, cu, cz, P)
            HG = self.HG(cols, n).T.reshape(E, self.neb, nav, nav)
            vc = self._vcols(cx, cu, cz)
            r0 = self.rows_per_edge - self.ne + blk * self.neb
            put_block(vc, vc, np.einsum("ei,eiab->eab", lamE[:, r0:r0 + self.neb], HG))
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

    def initial_guess(self, x0, u0=None, z0=None):
        """MPC.set_initial_guess (_mpc.py:955-973): every _x slot = x0 / x_scaling, every _u = u0 / u_scaling, every _z = z0 / z_scaling"""
        x = np.zeros(self.n_opt_x)
        x[:self.off_z].reshape(-1, self.nx)[:] = np.asarray(x0, float).ravel() / self.sx
        if z0 is not None:
            x[self.off_z:self.off_u].reshape(-1, self.nz)[:] = np.asarray(z0, float).ravel() / self.sz
        if u0 is not None:
            x[self.off_u:self.off_eps].reshape(-1, self.nu)[:] = np.asarray(u0, float).ravel() / self.su
        return x

    def scaling_vector(self):
        s = np.ones(self.n_opt_x)
        s[:self.off_z].reshape(-1, self.nx)[:] = self.sx
        s[self.off_z:self.off_u].reshape(-1, self.nz)[:] = self.sz
        s[self.off_u:self.off_eps].reshape(-1, self.nu)[:] = self.su
        return s

    def u0_of(self, x):
        return x[self.iu(0, 0):self.iu(0, 0) + self.nu] * self.su
