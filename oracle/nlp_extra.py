"""TEST INFRASTRUCTURE (oracle) - an NLP of oracle/nlp.py / nlp_dae.py with terms ADDED to its objective, as the reference's low-level route
does it:  `mpc.prepare_nlp();  mpc.nlp_obj += expr(opt_x, opt_p);  mpc.create_nlp()`  (/root/reference/do_mpc/optimizer.py:82-129 - the
property hands out the symbolic objective, the setter takes the extended one back; /root/reference/do_mpc/controller/_mpc.py:1326-1328 then
builds nlpsol from `{'f': nlp_obj, ...}` as it is).  The reference differentiates the one flat expression with CasADi; here the added part
is differentiated with sympy over exactly the variables it touches and added to f, grad f and the Hessian of the Lagrangian (times sigma)
of the wrapped NLP.  No knowledge of the scenario tree goes into this file: the product's grouping of such terms by tree node
(do_mpc_amd/nlp_route.py) is what it checks.  Only tests/ import it.
"""
import numpy as np
import scipy.sparse as sps
import sympy as sp


class AddedObjective:
    """nlp: OracleNLP / OracleNLPDae;  build(X, P) -> sympy expression, X / P: tuples of sympy symbols for opt_x / opt_p (scaled variables,
    the reference's flat order)."""

    def __init__(self, nlp, build):
        self._nlp = nlp
        X = sp.symbols("X0:%d" % nlp.n_opt_x)
        P = sp.symbols("P0:%d" % nlp.n_opt_p)
        ex = sp.sympify(build(X, P))
        ix = {s: i for i, s in enumerate(X)}
        ipp = {s: i for i, s in enumerate(P)}
        self._vx = sorted((s for s in ex.free_symbols if s in ix), key=lambda s: ix[s])
        self._vp = sorted((s for s in ex.free_symbols if s in ipp), key=lambda s: ipp[s])
        self._cx = np.array([ix[s] for s in self._vx], int)
        self._cp = np.array([ipp[s] for s in self._vp], int)
        args = list(self._vx) + list(self._vp)
        g = [sp.diff(ex, s) for s in self._vx]
        self._hij = [(a, b) for a in range(len(g)) for b in range(a, len(g)) if sp.diff(g[a], self._vx[b]) != 0]
        h = [sp.diff(g[a], self._vx[b]) for a, b in self._hij]
        self._fn = sp.lambdify(args, [ex] + g + h, modules="math", cse=True)

    def __getattr__(self, name):
        return getattr(self._nlp, name)

    def _eval(self, x, p):
        v = self._fn(*x[self._cx], *p[self._cp])
        n = len(self._cx)
        return float(v[0]), np.array(v[1:1 + n], float), np.array(v[1 + n:], float)

    def f(self, x, p):
        return self._nlp.f(x, p) + self._eval(x, p)[0]

    def grad(self, x, p):
        g = self._nlp.grad(x, p).copy()
        np.add.at(g, self._cx, self._eval(x, p)[1])
        return g

    def hess(self, x, p, sigma, lam):
        H = self._nlp.hess(x, p, sigma, lam)
        hv = self._eval(x, p)[2] * sigma
        r, c, v = [], [], []
        for (a, b), val in zip(self._hij, hv):
            r.append(self._cx[a]); c.append(self._cx[b]); v.append(val)
            if a != b:
                r.append(self._cx[b]); c.append(self._cx[a]); v.append(val)
        n = self._nlp.n_opt_x
        return (H + sps.csr_matrix((v, (r, c)), shape=(n, n))).tocsr()


class AddedConstraints:
    """`mpc.nlp_cons.append(expr); mpc.nlp_cons_lb.append(lb); mpc.nlp_cons_ub.append(ub)` between prepare_nlp() and create_nlp()
    (/root/reference/do_mpc/optimizer.py:131-215; create_nlp concatenates the blocks, optimizer.py:1086-1094 / _mpc.py:1303-1310): the rows
    follow the structured rows of the wrapped NLP, in the order they were appended.  build(X, P) -> list of sympy expressions."""

    def __init__(self, nlp, build, lb, ub):
        self._nlp = nlp
        X = sp.symbols("X0:%d" % nlp.n_opt_x)
        P = sp.symbols("P0:%d" % nlp.n_opt_p)
        rows = [sp.sympify(r) for r in build(X, P)]
        ix = {s: i for i, s in enumerate(X)}
        ipp = {s: i for i, s in enumerate(P)}
        free = set().union(*[r.free_symbols for r in rows])
        self._vx = sorted((s for s in free if s in ix), key=lambda s: ix[s])
        self._vp = sorted((s for s in free if s in ipp), key=lambda s: ipp[s])
        self._cx = np.array([ix[s] for s in self._vx], int)
        self._cp = np.array([ipp[s] for s in self._vp], int)
        args = list(self._vx) + list(self._vp)
        nv = len(self._vx)
        self._m = m = len(rows)
        J = [[sp.diff(r, s) for s in self._vx] for r in rows]
        self._jij = [(q, a) for q in range(m) for a in range(nv) if J[q][a] != 0]
        self._hij = [(q, a, b) for q in range(m) for a in range(nv) for b in range(a, nv) if sp.diff(J[q][a], self._vx[b]) != 0]
        outs = rows + [J[q][a] for q, a in self._jij] + [sp.diff(J[q][a], self._vx[b]) for q, a, b in self._hij]
        self._fn = sp.lambdify(args, outs, modules="math", cse=True)
        self.lbg = np.concatenate([nlp.lbg, np.asarray(lb, float).ravel()])
        self.ubg = np.concatenate([nlp.ubg, np.asarray(ub, float).ravel()])
        self.n_g = nlp.n_g + m

    def __getattr__(self, name):
        return getattr(self._nlp, name)

    def _eval(self, x, p):
        v = np.array(self._fn(*x[self._cx], *p[self._cp]), float)
        m, nj = self._m, len(self._jij)
        return v[:m], v[m:m + nj], v[m + nj:]

    def g(self, x, p):
        return np.concatenate([self._nlp.g(x, p), self._eval(x, p)[0]])

    def jac(self, x, p):
        jv = self._eval(x, p)[1]
        rows = [q for q, _ in self._jij]
        cols = [self._cx[a] for _, a in self._jij]
        Jx = sps.csr_matrix((jv, (rows, cols)), shape=(self._m, self._nlp.n_opt_x))
        return sps.vstack([self._nlp.jac(x, p), Jx]).tocsr()

    def hess(self, x, p, sigma, lam):
        n0 = self._nlp.n_g
        H = self._nlp.hess(x, p, sigma, lam[:n0])
        hv = self._eval(x, p)[2]
        r, c, v = [], [], []
        for (q, a, b), val in zip(self._hij, hv):
            t = lam[n0 + q] * val
            r.append(self._cx[a]); c.append(self._cx[b]); v.append(t)
            if a != b:
                r.append(self._cx[b]); c.append(self._cx[a]); v.append(t)
        n = self._nlp.n_opt_x
        return (H + sps.csr_matrix((v, (r, c)), shape=(n, n))).tocsr()
