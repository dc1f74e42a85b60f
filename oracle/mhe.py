"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's moving-horizon-estimation NLP.

Reference: /root/reference/do_mpc/estimator/_mhe.py
  variables    :1052-1060  opt_x = [_x (N+1, 1+M) | _z | _u (N) | _w (N) | _v (N) | _eps | _p_est]
  parameters   :1088-1094  opt_p = [_x_prev | _p_est_prev | _p_set | _tvp (N) | _y_meas (N)]
  objective    :1118-1127  arrival cost at `_x[0, -1]`, `_p_est`; :1190-1192 stage cost of (w_k, v_k, tvp_k, p)
  constraints  :1133-1188  per stage: collocation + continuity rows of the interval (optimizer.py:789-996), measurement rows
               h(x_{k+1}, u_k, tvp_k, p) + v_k - y_meas_k = 0, nl_cons rows at every stored point of the interval
               (nl_cons_check_colloc_points) - and, as the reference does (:1186-1188), the rows of the LAST evaluated
               point appended a second time
  bounds       :995-1028   (cons_check_colloc_points: every `_x` entry incl. the slots of stage 0)
  default objective :602-716  stage v'P_v v + w'P_w w, arrival (x0 - x_prev)'P_x(..) + (p - p_prev)'P_p(..)

One stage is written symbolically (sympy) over its own variables s_k = [x_k | slots of x_{k+1} | x_{k+1} | u_k | w_k | v_k | p_est];
constraint rows, Jacobian and the Hessian of (sigma l_k + lambda' c_k) are lambdified once and scattered into the global sparse
matrices.  Models without algebraic states, one finite element, all scalings 1 (the reference's only MHE example:
examples/rotating_oscillating_masses_mhe_mpc).  Interface = the one oracle/ipm.py uses.
"""
import numpy as np
import scipy.sparse as sps
import sympy as sp

from .nlp import collocation_coeffs


class OracleMHE:
    def __init__(self, case):
        self.case = c = case
        nx, nu = len(c["x"]), len(c["u"])
        nw, nv, ny = len(c.get("w", ())), len(c["v"]), len(c["meas"])
        self.nx, self.nu, self.nw, self.nv, self.ny = nx, nu, nw, nv, ny
        self.p_est, self.p_set = list(c["p_est"]), [q for q in c["p"] if q not in c["p_est"]]
        npe, nps = len(self.p_est), len(self.p_set)
        self.npe, self.nps, self.ntvp = npe, nps, len(c["tvp"])
        self.N = N = c["n_horizon"]
        self.discrete = c.get("model_type") == "discrete"            # (optimizer.py:820-824: ifcn = [alg, rhs], no stored points)
        self.deg = deg = 0 if self.discrete else c["collocation_deg"]
        assert c["collocation_ni"] == 1 and not c.get("z")
        self.nl_colloc = bool(c["nl_cons_check_colloc_points"]) and not self.discrete
        self.M = M = 0 if self.discrete else deg + 1
        self.h = c["t_step"]
        if not self.discrete:
            self.tau, self.C, self.D = collocation_coeffs(deg, c["collocation_type"])
        self.nl = c["nl_cons"]
        ne = len(self.nl)
        # ---- layouts
        self.off_u = (N + 1) * (M + 1) * nx
        self.off_w = self.off_u + N * nu
        self.off_v = self.off_w + N * nw
        # soft constraints (optimizer.py:543-585): one slack entry per soft row, `_eps` repeated n_eps times (_mhe.py:1046-1049:
        # N, or 1 with nl_cons_single_slack), stage k reads `_eps[min(k, n_eps - 1)]` (:1161); cost penalty * eps once per stage (:1196)
        self.soft = [i for i, q in enumerate(self.nl) if q.get("soft")]
        self.n_slack = len(self.soft)
        self.n_eps = 1 if c.get("nl_cons_single_slack") else N
        self.off_eps = self.off_v + N * nv
        self.off_p = self.off_eps + self.n_eps * self.n_slack
        self.n_opt_x = self.off_p + npe
        self.po_pprev = nx
        self.po_pset = nx + npe
        self.po_tvp = self.po_pset + nps
        self.po_y = self.po_tvp + N * self.ntvp
        self.n_opt_p = self.po_y + N * ny
        self.n_eval = (M if self.nl_colloc else 1) + 1          # evaluations of the nl_cons rows per stage (the last one twice, :1186-1188)
        self.rows_stage = M * nx + nx + ny + self.n_eval * ne
        self.n_g = N * self.rows_stage
        self._build_stage()
        self._build_bounds()

    def ix(self, k, c):
        return (k * (self.M + 1) + c) * self.nx

    # ------------------------------------------------------------------ one stage, symbolically
    def _build_stage(self):
        c = self.case
        nx, nu, nw, nv, ny, M, deg, npe = self.nx, self.nu, self.nw, self.nv, self.ny, self.M, self.deg, self.npe
        xs = [sp.symbols(f"xa0:{nx}")] + [sp.symbols(f"xs{i}_0:{nx}") for i in range(M)] + [sp.symbols(f"xb0:{nx}")]
        us, ws, vs = sp.symbols(f"uu0:{nu}"), sp.symbols(f"ww0:{nw}") if nw else (), sp.symbols(f"vv0:{nv}")
        pe = sp.symbols(f"pe0:{npe}") if npe else ()
        es = sp.symbols(f"ee0:{self.n_slack}") if self.n_slack else ()
        pset = sp.symbols(f"ps0:{self.nps}") if self.nps else ()
        tv = sp.symbols(f"tv0:{self.ntvp}") if self.ntvp else ()
        ym = sp.symbols(f"ym0:{ny}")
        pmap = {q: pe[self.p_est.index(q)] for q in self.p_est}
        pmap.update({q: pset[self.p_set.index(q)] for q in self.p_set})
        tmap = {c["tvp"][i]: tv[i] for i in range(self.ntvp)}

        def at(expr, xv, uv=us, wv=ws, vv=vs):
            sub = {c["x"][i]: xv[i] for i in range(nx)}
            sub.update({c["u"][i]: uv[i] for i in range(nu)})
            sub.update({c["w"][i]: wv[i] for i in range(nw)} if nw else {})
            sub.update({c["v"][i]: vv[i] for i in range(nv)})
            sub.update(pmap); sub.update(tmap)
            return sp.sympify(expr).subs(sub)

        rows = []
        xb = xs[M + 1]
        if self.discrete:
            for a in range(nx):                                         # x+ = f(x, u, w, p)  (_mhe.py:1153-1155 with xf = rhs)
                rows.append(at(c["rhs"][a], xs[0]) - xb[a])
        else:
            pts = [xs[0]] + [xs[1 + r] for r in range(deg)]             # point 0 = x_k, points 1..deg = the collocation slots
            for j in range(1, deg + 1):                                 # collocation rows  h f(x_j) - sum_r C[r, j] x_r
                for a in range(nx):
                    rows.append(self.h * at(c["rhs"][a], pts[j]) - sum(self.C[r, j] * pts[r][a] for r in range(deg + 1)))
            xe = xs[M]                                                  # end-of-element slot
            for a in range(nx):
                rows.append(xe[a] - sum(self.D[r] * pts[r][a] for r in range(deg + 1)))
            for a in range(nx):                                         # continuity
                rows.append(xe[a] - xb[a])
        for i in range(ny):                                             # measurement rows (meas_fun includes the noise v)
            rows.append(at(c["meas"][i], xb) - ym[i])
        # nl_cons rows: at every stored point of the interval (`_x[k+1, i]`) or at the state `_x[k, -1]`; then the rows of the last
        # evaluated point once more (_mhe.py:1186-1188)
        pts_nl = [xs[1 + b] for b in range(M)] if self.nl_colloc else [xs[0]]
        for xv in pts_nl + [pts_nl[-1]]:
            for i, ncn in enumerate(self.nl):
                rows.append(at(ncn["expr"], xv) - (es[self.soft.index(i)] if i in self.soft else 0))
        lk = at(c["stage_cost"], xb) + sum(self.nl[i]["penalty"] * es[q] for q, i in enumerate(self.soft))
        svars = [s for blk in xs for s in blk] + list(us) + list(ws) + list(vs) + list(es) + list(pe)
        self.ns = len(svars)
        lam = sp.symbols(f"lm0:{len(rows)}")
        sig = sp.Symbol("sg")
        args = svars + list(pset) + list(tv) + list(ym)
        J = [[sp.diff(r, s) for s in svars] for r in rows]
        L = sig * lk + sum(l * r for l, r in zip(lam, rows))
        gL = [sp.diff(L, s) for s in svars]
        H = [[sp.diff(gL[i], svars[j]) for j in range(self.ns)] for i in range(self.ns)]
        self.Jnz = [(i, j) for i in range(len(rows)) for j in range(self.ns) if J[i][j] != 0]
        self.Hnz = [(i, j) for i in range(self.ns) for j in range(self.ns) if H[i][j] != 0]
        mk = lambda ex, ar: sp.lambdify(ar, ex, modules="numpy", cse=True)     # noqa: E731
        self.f_rows = mk(rows, args)
        self.f_J = mk([J[i][j] for i, j in self.Jnz], args)
        self.f_H = mk([H[i][j] for i, j in self.Hnz], args + [sig] + list(lam))
        self.f_l = mk([lk], args)
        self.f_gl = mk([sp.diff(lk, s) for s in svars], args)
        # arrival cost over (x_0, p_est; x_prev, p_prev, p_set)
        x0 = sp.symbols(f"xz0:{nx}")
        xp, pp = sp.symbols(f"xp0:{nx}"), sp.symbols(f"pp0:{npe}") if npe else ()
        amap = {c["x"][i]: x0[i] for i in range(nx)}
        amap.update({c["x_prev"][i]: xp[i] for i in range(nx)})
        amap.update({c["p_est_prev"][i]: pp[i] for i in range(npe)})
        amap.update(pmap)
        A = sp.sympify(c["arrival_cost"]).subs(amap)
        av = list(x0) + list(pe)
        aargs = av + list(xp) + list(pp) + list(pset)
        self.f_a = mk([A], aargs)
        self.f_ga = mk([sp.diff(A, s) for s in av], aargs)
        self.f_Ha = mk([sp.diff(A, a, b) for a in av for b in av], aargs)

    # ------------------------------------------------------------------ index maps
    def _svars(self, k):
        nx, M = self.nx, self.M
        idx = [self.ix(k, M) + np.arange(nx)]
        idx += [self.ix(k + 1, s) + np.arange(nx) for s in range(M)]
        idx += [self.ix(k + 1, M) + np.arange(nx)]
        idx += [self.off_u + k * self.nu + np.arange(self.nu), self.off_w + k * self.nw + np.arange(self.nw),
                self.off_v + k * self.nv + np.arange(self.nv),
                self.off_eps + min(k, self.n_eps - 1) * self.n_slack + np.arange(self.n_slack), self.off_p + np.arange(self.npe)]
        return np.concatenate(idx).astype(int)

    def _args(self, x, p, k):
        s = x[self._svars(k)]
        return list(s) + list(p[self.po_pset:self.po_tvp]) + list(p[self.po_tvp + k * self.ntvp:self.po_tvp + (k + 1) * self.ntvp]) \
            + list(p[self.po_y + k * self.ny:self.po_y + (k + 1) * self.ny])

    def _aargs(self, x, p):
        return list(x[self.ix(0, self.M):self.ix(0, self.M) + self.nx]) + list(x[self.off_p:]) + list(p[:self.po_pset]) \
            + list(p[self.po_pset:self.po_tvp])

    # ------------------------------------------------------------------ bounds
    def _build_bounds(self):
        c = self.case
        lb, ub = -np.inf * np.ones(self.n_opt_x), np.inf * np.ones(self.n_opt_x)
        XL, XU = lb[:self.off_u].reshape(self.N + 1, self.M + 1, self.nx), ub[:self.off_u].reshape(self.N + 1, self.M + 1, self.nx)
        if c.get("cons_check_colloc_points", True):
            XL[:], XU[:] = c["x_lb"], c["x_ub"]
        else:
            XL[1:self.N, -1], XU[1:self.N, -1] = c["x_lb"], c["x_ub"]
        lb[self.off_u:self.off_w].reshape(-1, self.nu)[:] = c["u_lb"]
        ub[self.off_u:self.off_w].reshape(-1, self.nu)[:] = c["u_ub"]
        if self.n_slack:
            lb[self.off_eps:self.off_p] = 0.0
            ub[self.off_eps:self.off_p].reshape(-1, self.n_slack)[:] = [self.nl[i].get("max_violation", np.inf) for i in self.soft]
        if self.npe:
            lb[self.off_p:], ub[self.off_p:] = c.get("p_est_lb", -np.inf), c.get("p_est_ub", np.inf)
        self.lbx, self.ubx = lb, ub
        lbg, ubg = np.zeros(self.n_g), np.zeros(self.n_g)
        ne = len(self.nl)
        if ne:
            G = lbg.reshape(self.N, self.rows_stage)
            G[:, self.rows_stage - self.n_eval * ne:] = -np.inf
            U = ubg.reshape(self.N, self.rows_stage)
            U[:, self.rows_stage - self.n_eval * ne:] = np.tile([q["ub"] for q in self.nl], self.n_eval)
        self.lbg, self.ubg = lbg, ubg

    # ------------------------------------------------------------------ functions
    def f(self, x, p):
        v = float(np.asarray(self.f_a(*self._aargs(x, p))[0]))
        for k in range(self.N):
            v += float(np.asarray(self.f_l(*self._args(x, p, k))[0]))
        return v

    def grad(self, x, p):
        g = np.zeros(self.n_opt_x)
        ga = np.array([float(np.asarray(q)) for q in self.f_ga(*self._aargs(x, p))])
        g[self.ix(0, self.M):self.ix(0, self.M) + self.nx] += ga[:self.nx]
        g[self.off_p:] += ga[self.nx:]
        for k in range(self.N):
            gl = np.array([float(np.asarray(q)) for q in self.f_gl(*self._args(x, p, k))])
            np.add.at(g, self._svars(k), gl)
        return g

    def g(self, x, p):
        out = np.zeros(self.n_g)
        for k in range(self.N):
            out[k * self.rows_stage:(k + 1) * self.rows_stage] = [float(np.asarray(q)) for q in self.f_rows(*self._args(x, p, k))]
        return out

    def jac(self, x, p):
        R, Cc, V = [], [], []
        ji = np.array([i for i, _ in self.Jnz]); jj = np.array([j for _, j in self.Jnz])
        for k in range(self.N):
            sv = self._svars(k)
            vals = np.array([float(np.asarray(q)) for q in self.f_J(*self._args(x, p, k))])
            R.append(k * self.rows_stage + ji); Cc.append(sv[jj]); V.append(vals)
        return sps.csr_matrix((np.concatenate(V), (np.concatenate(R), np.concatenate(Cc))), shape=(self.n_g, self.n_opt_x))

    def hess(self, x, p, sigma, lam):
        R, Cc, V = [], [], []
        hi = np.array([i for i, _ in self.Hnz]); hj = np.array([j for _, j in self.Hnz])
        for k in range(self.N):
            sv = self._svars(k)
            lk = lam[k * self.rows_stage:(k + 1) * self.rows_stage]
            if len(self.Hnz):
                vals = np.array([float(np.asarray(q)) for q in self.f_H(*(self._args(x, p, k) + [sigma] + list(lk)))])
                R.append(sv[hi]); Cc.append(sv[hj]); V.append(vals)
        na = self.nx + self.npe
        Ha = sigma * np.array([float(np.asarray(q)) for q in self.f_Ha(*self._aargs(x, p))]).reshape(na, na)
        av = np.concatenate([self.ix(0, self.M) + np.arange(self.nx), self.off_p + np.arange(self.npe)]).astype(int)
        R.append(np.repeat(av, na)); Cc.append(np.tile(av, na)); V.append(Ha.ravel())
        return sps.csr_matrix((np.concatenate(V), (np.concatenate(R), np.concatenate(Cc))), shape=(self.n_opt_x, self.n_opt_x))

    # ------------------------------------------------------------------ protocol helpers
    def initial_guess(self, x0, u0=None, p_est0=None):
        """MHE.set_initial_guess (_mhe.py:864-883): every `_x` entry = x0, every `_u` = u0, `_p_est` = p_est0, noise 0"""
        v = np.zeros(self.n_opt_x)
        v[:self.off_u].reshape(-1, self.nx)[:] = x0
        if u0 is not None:
            v[self.off_u:self.off_w].reshape(-1, self.nu)[:] = u0
        if p_est0 is not None:
            v[self.off_p:] = p_est0
        return v

    def x_last(self, x):
        return x[self.ix(self.N, self.M):self.ix(self.N, self.M) + self.nx]
