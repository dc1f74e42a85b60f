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
