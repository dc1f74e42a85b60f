"""Generic sparse primal-dual interior-point solver (IPOPT's published algorithm).

TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.

The reference delegates the solve to a third-party binary that is absent here:
  casadi.nlpsol('S','ipopt', nlp, opts)    /root/reference/do_mpc/controller/_mpc.py:1326-1328
  r = self.S(x0=,lbx=,ubx=,lbg=,ubg=,p=)   /root/reference/do_mpc/optimizer.py:754-778
(PyPI `casadi>=3.6.0`, /root/reference/requirements.txt:1, bundling IPOPT 3.14 + MUMPS).
This file restates IPOPT's algorithm from its publication
  A. Waechter, L. T. Biegler, "On the implementation of an interior-point filter
  line-search algorithm for large-scale nonlinear programming", Math. Prog. 106 (2006)
with IPOPT 3.14's documented default options (names in comments).  It works on the
flat NLP (oracle/nlp.py) with a general sparse LU of the augmented system - no
stage structure - so it is an independent check of the product's Riccati path.
Parity is pinned on the reference's golden vectors (tests/golden/*.npz), see
tests/test_oracle_golden.py.

Deviations from IPOPT (documented, none changes the limit point of a convex problem):
  * inertia is not available from scipy's sparse LU: by default the inertia-correction loop uses the
    curvature test d'(W+Sigma+delta I)d >= kappa |d|^2 (IPOPT option neg_curv_test)
    instead of counting negative pivots.  opts["inertia"] = "ldl" counts them (dense Bunch-Kaufman
    LDL' of the augmented matrix, scipy.linalg.ldl: n positive / m negative eigenvalues required, as
    IPOPT does with MUMPS) - O(n^3), for the small non-convex cases of the test-suite;
  * no restoration phase: if the backtracking line search hits alpha_min the last
    trial step is taken and the filter is reset (counted in stats['n_ls_fail']).
Beyond the paper, from IPOPT's implementation (both default-on there): the damping of one-sided bounds (kappa_d) and the watchdog
procedure of the line search (watchdog_shortened_iter_trigger / watchdog_trial_iter_max, see DEFAULTS).
"""
import time

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla

DEFAULTS = dict(
    tol=1e-8, max_iter=3000, dual_inf_tol=1.0, constr_viol_tol=1e-4, compl_inf_tol=1e-4,
    acceptable_tol=1e-6, acceptable_iter=15,
    mu_init=0.1, mu_target=0.0, kappa_mu=0.2, theta_mu=1.5, kappa_eps=10.0, tau_min=0.99,
    bound_push=0.01, bound_frac=0.01, bound_relax_factor=1e-8, bound_mult_init_val=1.0,
    constr_mult_init_max=1000.0, s_max=100.0, kappa_sigma=1e10,
    eta_phi=1e-8, gamma_theta=1e-5, gamma_phi=1e-8, delta=1.0, s_theta=1.1, s_phi=2.3,
    theta_max_fact=1e4, theta_min_fact=1e-4, gamma_alpha=0.05, max_soc=4, kappa_soc=0.99,
    delta_w_min=1e-20, delta_w_0=1e-4, delta_w_max=1e40, kappa_w_minus=1.0 / 3.0,
    kappa_w_plus=8.0, kappa_w_plus_bar=100.0, delta_c_bar=1e-8, kappa_c=0.25,
    ls_mult_init=True, inertia="curvature",
    fast=False,         # bench.py's cpu_baseline: same algorithm and control flow, cheaper linear algebra - see _FastKKT below
    kappa_d=1e-5,       # linear damping of the barrier for variables with ONE bound (section 3.7 of the paper)
    nlp_scaling_max_gradient=100.0, nlp_scaling_min_value=1e-8, obj_scaling=True, con_scaling=True,
    # IPOPT's watchdog procedure (IpBacktrackingLineSearch: options watchdog_shortened_iter_trigger = 10, watchdog_trial_iter_max = 3;
    # not in the 2006 paper): after that many consecutive iterations whose step was shortened by the backtracking, up to 3 full steps
    # are taken without asking the filter, each tested against the point where the watchdog started; none acceptable: back to that
    # point and its direction, regular backtracking from the second trial step size.  0 = off.
    watchdog_shortened_iter_trigger=10, watchdog_trial_iter_max=3,
)


class Result(dict):
    pass


def _n_negative(K):
    """Number of negative eigenvalues of the symmetric matrix K (Sylvester: those of the block-diagonal factor of its
    Bunch-Kaufman LDL' factorisation); -1 if a pivot block is exactly singular (MUMPS, as IPOPT runs it, does not
    declare small pivots singular either)."""
    import scipy.linalg as sla
    _, d, _ = sla.ldl(K.toarray(), lower=True)
    n, neg, i = d.shape[0], 0, 0
    while i < n:
        if i + 1 < n and d[i + 1, i] != 0.0:
            ev = np.linalg.eigvalsh(d[i:i + 2, i:i + 2])
            i += 2
        else:
            ev = d[i:i + 1, i]
            i += 1
        if np.any(ev == 0.0):
            return -1
        neg += int(np.sum(ev < 0))
    return neg


def _n_negative_sparse(Hreg, A, delta_c):
    """Exact number of negative eigenvalues of K = [[Hreg, A'], [A, -delta_c I]] from a sparse LDL' WITHOUT pivoting: the variables in a
    reverse Cuthill-McKee order of the pattern of Hreg + A'A, every constraint row directly behind the last of its variables - its pivot is
    then the (non-zero) Schur complement -a (..)^-1 a' instead of the structural zero - and SuperLU in its natural order with diagonal
    pivots only (symmetric mode, diag_pivot_thresh = 0).  Without a row interchange K = L D L' with D = diag(U), and Sylvester's law of
    inertia gives the count (0.1 s for the 15 310 rows of industrial_poly; the dense Bunch-Kaufman count takes 20 - 45 s).  Returns -1 if
    SuperLU had to interchange rows after all (the caller falls back to the dense count)."""
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    nv, m = Hreg.shape[0], A.shape[0]
    Ac = A.tocsr()
    G = (abs(Hreg) + abs(Ac.T) @ abs(Ac)).tocsr()
    pv = np.asarray(reverse_cuthill_mckee(G, symmetric_mode=True))
    pos = np.empty(nv, dtype=np.int64)
    pos[pv] = np.arange(nv)
    last = np.full(m, -1, dtype=np.int64)
    rows = np.repeat(np.arange(m), np.diff(Ac.indptr))
    np.maximum.at(last, rows, pos[Ac.indices])
    order = np.lexsort((np.arange(nv + m), np.concatenate([2 * pos, 2 * last + 1])))
    K = sps.bmat([[Hreg, Ac.T], [Ac, -delta_c * sps.identity(m) if delta_c > 0 else None]], format="csc")
    Kp = K[order][:, order].tocsc()
    try:
        lu = spla.splu(Kp, permc_spec="NATURAL", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    except RuntimeError:
        return -1
    ident = np.arange(nv + m)
    if not (np.array_equal(lu.perm_r, ident) and np.array_equal(lu.perm_c, ident)):
        return -1
    d = lu.U.diagonal()
    if not np.all(np.isfinite(d)) or np.any(d == 0.0):
        return -1
    return int(np.sum(d < 0))


class _FastKKT:
    """opts["fast"] (the CPU baseline of bench.py; the tests run without it): the augmented system of every iteration with
      * a STRUCTURAL singularity test in place of the two factorisations that IPOPT / the plain path spend on finding out that the
        system is singular at delta_w = 0 - a variable that appears in no constraint row, no Hessian entry and has no bound is a
        zero row and column for every delta_c, so the verdict (and with it the delta_w sequence, the iterates) is the same;
      * one reverse Cuthill-McKee ordering of the (constant) sparsity pattern, computed once per solve, and SuperLU in its
        natural column order on the permuted matrix instead of a COLAMD ordering per factorisation (3x fewer flops on the
        stage-structured KKT matrices of do-mpc's NLPs);
      * the sparse assembly of K from cached index maps instead of scipy.sparse.bmat per attempt.
    Same right-hand sides, same iterative refinement; the solutions differ from the plain path at rounding level."""

    def __init__(self):
        self.key = None

    def structurally_singular(self, W, A, sigma, delta_w):
        if delta_w != 0.0:
            return False
        empty = (np.diff(A.tocsc().indptr) == 0) & (np.diff(W.tocsr().indptr) == 0) & (np.diff(W.tocsc().indptr) == 0)
        return bool(np.any(empty & (sigma == 0.0)))

    def factor(self, W, A, diag, delta_c, m):
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        nv = W.shape[0]
        Wc, Ac = W.tocoo(), A.tocoo()
        key = (Wc.nnz, Ac.nnz, nv, m, delta_c > 0)
        if self.key != key or not (np.array_equal(Wc.row, self.wr) and np.array_equal(Wc.col, self.wc)
                                   and np.array_equal(Ac.row, self.ar) and np.array_equal(Ac.col, self.ac)):
            self.key, self.wr, self.wc, self.ar, self.ac = key, Wc.row.copy(), Wc.col.copy(), Ac.row.copy(), Ac.col.copy()
            rows = np.concatenate([Wc.row, np.arange(nv), Ac.col, Ac.row + nv] + ([np.arange(m) + nv] if delta_c > 0 else []))
            cols = np.concatenate([Wc.col, np.arange(nv), Ac.row + nv, Ac.col] + ([np.arange(m) + nv] if delta_c > 0 else []))
            pat = sps.csr_matrix((np.ones(rows.size), (rows, cols)), shape=(nv + m, nv + m))
            self.perm = np.asarray(reverse_cuthill_mckee(pat, symmetric_mode=True))
            inv = np.empty_like(self.perm)
            inv[self.perm] = np.arange(self.perm.size)
            self.inv = inv
            self.rows, self.cols = inv[rows], inv[cols]
        data = np.concatenate([Wc.data, diag, Ac.data, Ac.data] + ([np.full(m, -delta_c)] if delta_c > 0 else []))
        K = sps.csc_matrix((data, (self.rows, self.cols)), shape=(nv + m, nv + m))      # (duplicates are summed)
        lu = spla.splu(K, permc_spec="NATURAL")
        perm, inv = self.perm, self.inv

        class _LU:
            def solve(self_, b):
                return lu.solve(b[perm])[inv]

            def matvec(self_, v):
                return (K @ v[perm])[inv]
        return _LU()


def solve(nlp, x0, p, lam_x0=None, lam_g0=None, opts=None, lbx=None, ubx=None, lbg=None, ubg=None, trace=None):
    """Returns dict with x, f, g, lam_g, lam_x and stats (CasADi sign convention
    L = f + lam_g' g + lam_x' x, /root/reference/do_mpc/differentiator/_nlpdifferentiator.py:287)."""
    o = dict(DEFAULTS)
    if opts:
        o.update(opts)
    t_start = time.perf_counter()
    n = nlp.n_opt_x
    lbx = nlp.lbx if lbx is None else lbx
    ubx = nlp.ubx if ubx is None else ubx
    lbg = nlp.lbg if lbg is None else lbg
    ubg = nlp.ubg if ubg is None else ubg
    eq = lbg == ubg
    ineq = ~eq
    m_e, m_i = int(eq.sum()), int(ineq.sum())
    m = m_e + m_i
    idx_e, idx_i = np.where(eq)[0], np.where(ineq)[0]

    # ---- variables v = [x ; s]; bounds with relaxation (bound_relax_factor)
    vl = np.concatenate([lbx, lbg[idx_i]]).astype(float)
    vu = np.concatenate([ubx, ubg[idx_i]]).astype(float)
    relax = o["bound_relax_factor"]
    has_l, has_u = np.isfinite(vl), np.isfinite(vu)
    vl[has_l] -= np.minimum(o["constr_viol_tol"], relax * np.maximum(1.0, np.abs(vl[has_l])))
    vu[has_u] += np.minimum(o["constr_viol_tol"], relax * np.maximum(1.0, np.abs(vu[has_u])))
    nv = n + m_i

    def push(v, l, u, hl, hu):
        v = v.copy()
        both = hl & hu
        pl = np.zeros_like(v)
        pu = np.zeros_like(v)
        pl[hl] = o["bound_push"] * np.maximum(1.0, np.abs(l[hl]))
        pu[hu] = o["bound_push"] * np.maximum(1.0, np.abs(u[hu]))
        pl[both] = np.minimum(pl[both], o["bound_frac"] * (u[both] - l[both]))
        pu[both] = np.minimum(pu[both], o["bound_frac"] * (u[both] - l[both]))
        v[hl] = np.maximum(v[hl], l[hl] + pl[hl])
        v[hu] = np.minimum(v[hu], u[hu] - pu[hu])
        return v

    x = push(np.asarray(x0, float).copy(), vl[:n], vu[:n], has_l[:n], has_u[:n])

    # ---- NLP scaling (gradient based, nlp_scaling_method default)
    g0 = nlp.grad(x, p)
    J0 = nlp.jac(x, p)
    gmax = o["nlp_scaling_max_gradient"]
    sf = 1.0
    if o["obj_scaling"]:
        gn = np.max(np.abs(g0)) if n else 0.0
        if gn > gmax:
            sf = max(gmax / gn, o["nlp_scaling_min_value"])
    sg = np.ones(len(lbg))
    if o["con_scaling"] and J0.nnz:
        rn = np.asarray(abs(J0).max(axis=1).todense()).ravel()
        big = rn > gmax
        sg[big] = np.maximum(gmax / rn[big], o["nlp_scaling_min_value"])
    # scaled inequality bounds
    vl[n:] *= sg[idx_i]
    vu[n:] *= sg[idx_i]
    Sg = sps.diags(sg)
    ce_rhs = lbg[idx_e] * sg[idx_e]

    n_eval = dict(f=0, g=0, jac=0, hess=0)

    def eval_fg(xx):
        n_eval["f"] += 1
        n_eval["g"] += 1
        return sf * nlp.f(xx, p), sg * nlp.g(xx, p)

    def cons(gv, s):
        c = np.empty(m)
        c[:m_e] = gv[idx_e] - ce_rhs
        c[m_e:] = gv[idx_i] - s
        return c

    fval, gval = eval_fg(x)
    s = push(gval[idx_i].copy(), vl[n:], vu[n:], has_l[n:], has_u[n:])
    v = np.concatenate([x, s])
    zl = np.where(has_l, o["bound_mult_init_val"], 0.0)
    zu = np.where(has_u, o["bound_mult_init_val"], 0.0)

    def jac_full(xx):
        n_eval["jac"] += 1
        J = (Sg @ nlp.jac(xx, p)).tocsr()
        Je, Ji = J[idx_e], J[idx_i]
        top = sps.hstack([Je, sps.csr_matrix((m_e, m_i))]) if m_e else sps.csr_matrix((0, nv))
        bot = sps.hstack([Ji, -sps.identity(m_i)]) if m_i else sps.csr_matrix((0, nv))
        return sps.vstack([top, bot]).tocsr()

    def grad_full(xx):
        gfull = np.zeros(nv)
        gfull[:n] = sf * nlp.grad(xx, p)
        return gfull

    A = jac_full(x)
    gf = grad_full(x)

    # ---- least-square multiplier estimate
    def ls_multipliers():
        K = sps.bmat([[sps.identity(nv), A.T], [A, None]], format="csc")
        rhs = np.concatenate([-(gf - zl + zu), np.zeros(m)])
        try:
            sol = spla.splu(K).solve(rhs)
            y = sol[nv:]
            if not np.all(np.isfinite(y)) or np.max(np.abs(y), initial=0.0) > o["constr_mult_init_max"]:
                y = np.zeros(m)
        except RuntimeError:
            y = np.zeros(m)
        return y

    y = ls_multipliers() if (m and o["ls_mult_init"]) else np.zeros(m)
    mu = o["mu_init"]
    tau = max(o["tau_min"], 1.0 - mu)
    # IPOPT's monotone update floors mu at min(tol, compl_inf_tol)/(barrier_tol_factor+1)
    # (= 1e-8/11 = 9.0909e-10 with defaults).  Visible in the reference's goldens: the CSTR
    # slack eps sits at -1e-8 + 9.0909e-12 = lower bound(relaxed) + mu/penalty.
    mu_min = min(o["tol"], o["compl_inf_tol"] * sf) / (o["kappa_eps"] + 1.0)

    def lam_unscaled(yy):
        lam = np.zeros(len(lbg))
        lam[idx_e] = yy[:m_e]
        lam[idx_i] = yy[m_e:]
        return lam * sg / sf

    def hess_full(xx, yy):
        n_eval["hess"] += 1
        lam = np.zeros(len(lbg))
        lam[idx_e] = yy[:m_e]
        lam[idx_i] = yy[m_e:]
        H = nlp.hess(xx, p, sf, lam * sg)
        return sps.block_diag([H, sps.csr_matrix((m_i, m_i))], format="csr") if m_i else H.tocsr()

    only_l, only_u = has_l & ~has_u, has_u & ~has_l

    def damp(mu_):
        """gradient of the damping term kappa_d mu (sum_{lower only}(v - l) + sum_{upper only}(u - v))"""
        return o["kappa_d"] * mu_ * (only_l.astype(float) - only_u.astype(float))

    def err(mu_, v_, y_, zl_, zu_, gf_, A_, c_):
        dl = np.where(has_l, v_ - vl, 1.0)
        du = np.where(has_u, vu - v_, 1.0)
        rd = gf_ + A_.T @ y_ - zl_ + zu_ + damp(mu_)
        sd = max(o["s_max"], (np.abs(y_).sum() + np.abs(zl_).sum() + np.abs(zu_).sum()) / max(1, m + has_l.sum() + has_u.sum())) / o["s_max"]
        sc = max(o["s_max"], (np.abs(zl_).sum() + np.abs(zu_).sum()) / max(1, has_l.sum() + has_u.sum())) / o["s_max"]
        e_d = np.max(np.abs(rd), initial=0.0)
        e_p = np.max(np.abs(c_), initial=0.0)
        e_c = max(np.max(np.abs(dl * zl_ - mu_)[has_l], initial=0.0), np.max(np.abs(du * zu_ - mu_)[has_u], initial=0.0))
        return max(e_d / sd, e_p, e_c / sc), e_d, e_p, e_c

    def barrier(fv, v_):
        return fv - mu * (np.log((v_ - vl)[has_l]).sum() + np.log((vu - v_)[has_u]).sum()) \
            + o["kappa_d"] * mu * ((v_ - vl)[only_l].sum() + (vu - v_)[only_u].sum())

    fast = _FastKKT() if o["fast"] else None
    c = cons(gval, s)
    filt = []
    theta0 = np.abs(c).sum()
    theta_max = o["theta_max_fact"] * max(1.0, theta0)
    theta_min = o["theta_min_fact"] * max(1.0, theta0)
    delta_w_last = 0.0
    stats = dict(n_ls_fail=0, n_reg=0, n_soc=0, iters=[], n_ldl_checks=0, n_proxy_false_rejections=0)
    status = "Maximum_Iterations_Exceeded"
    success = False
    acc_count = 0
    it = 0
    wd_count, in_wd, wd_iter, wd, wd_restart = 0, False, 0, None, None
    stats["n_watchdog"] = 0
    while True:
        if wd_restart is None:
            E0, e_d, e_p, e_c = err(0.0, v, y, zl, zu, gf, A, c)
            if trace is not None:
                trace.append(dict(it=it, mu=mu, E0=E0, inf_du=e_d, inf_pr=e_p, compl=e_c, f=fval / sf, x=v[:n].copy(), y=y.copy(), zl=zl[:n].copy(), zu=zu[:n].copy(), sf=sf, delta_w_last=delta_w_last))
            # unscaled acceptance thresholds are checked on scaled quantities here (scaling is mild)
            if E0 <= o["tol"] and e_d <= o["dual_inf_tol"] and e_p <= o["constr_viol_tol"] and e_c <= o["compl_inf_tol"]:
                status, success = "Solve_Succeeded", True
                break
            if E0 <= o["acceptable_tol"]:
                acc_count += 1
                if acc_count >= o["acceptable_iter"]:
                    status, success = "Solved_To_Acceptable_Level", True
                    break
            else:
                acc_count = 0
            if it >= o["max_iter"]:
                break
        if wd_restart is not None:
            # the watchdog gave up: back at the point where it started, with the direction computed there
            (v, y, zl, zu, fval, gval, c, A, gf, mu, tau, filt, delta_w, delta_w_last, lu, rx, dv, dy, dzl, dzu, a_max, a_z, dl, du) = wd_restart
            wd_restart = None
            skip_first = True
        else:
            skip_first = False
            # ---- barrier update
            while True:
                Emu = err(mu, v, y, zl, zu, gf, A, c)[0]
                if Emu <= o["kappa_eps"] * mu and mu > mu_min:
                    mu = max(mu_min, min(o["kappa_mu"] * mu, mu ** o["theta_mu"]))
                    tau = max(o["tau_min"], 1.0 - mu)
                    filt = []
                    in_wd, wd_count = False, 0          # (a new barrier problem: reference point and filter of the watchdog are void)
                else:
                    break
            # ---- search direction
            W = hess_full(v[:n], y)
            dl = np.where(has_l, v - vl, 1.0)
            du = np.where(has_u, vu - v, 1.0)
            sigma = np.where(has_l, zl / dl, 0.0) + np.where(has_u, zu / du, 0.0)
            rx = gf + A.T @ y - np.where(has_l, mu / dl, 0.0) + np.where(has_u, mu / du, 0.0) + damp(mu)
            rhs = -np.concatenate([rx, c])
            delta_w, delta_c = 0.0, 0.0
            first_try = True
            tried_c = False
            while True:
                Hreg = W + sps.diags(sigma + delta_w)
                ok = True
                try:
                    if fast is not None:
                        if fast.structurally_singular(W, A, sigma, delta_w):
                            raise RuntimeError("structurally singular")
                        lu = fast.factor(W, A, sigma + delta_w, delta_c, m)
                        kmul = lu.matvec
                    else:
                        K = sps.bmat([[Hreg, A.T], [A, -delta_c * sps.identity(m) if delta_c > 0 else None]], format="csc")
                        lu = spla.splu(K)
                        kmul = K.__matmul__
                    sol = lu.solve(rhs)
                    # one step of iterative refinement (IPOPT: min_refinement_steps = 1); opts["refine"] = 0 leaves it out - the
                    # product's structured solve has none, which can delay the 1e-8 termination test by an iteration or two
                    for _ in range(int(o.get("refine", 1))):
                        sol += lu.solve(rhs - kmul(sol))
                    ok = np.all(np.isfinite(sol))
                except RuntimeError:
                    # singular system (IpPDPerturbationHandler::PerturbForSingularity while the kind of degeneracy is unknown):
                    # first delta_c > 0 with delta_w = 0; if that is singular too, delta_c back to 0 and delta_w from the
                    # wrong-inertia rule.  (A system that is singular because of zero rows AND columns of the Hessian block -
                    # the unused variables of the do-mpc NLP - always ends in the second case: delta_c = 0.)
                    ok = False
                    if o.get("ipopt_delta_c_sequence", True):
                        if not tried_c and delta_w == 0.0:
                            tried_c = True
                            delta_c = o["delta_c_bar"] * mu ** o["kappa_c"]
                            continue
                        delta_c = 0.0
                    elif delta_c == 0.0:
                        delta_c = o["delta_c_bar"] * mu ** o["kappa_c"]
                if ok and o["inertia"] == "ldl":
                    if fast is not None:
                        K = sps.bmat([[Hreg, A.T], [A, -delta_c * sps.identity(m) if delta_c > 0 else None]], format="csc")
                    if _n_negative(K) == m:
                        break
                elif ok:
                    # (variables that appear in no constraint and no Hessian entry - the unused slots of the do-mpc NLP - are
                    #  decoupled 1x1 blocks with the pivot sigma + delta_w > 0: they cannot spoil the inertia, but with
                    #  delta_w ~ 1e-12 their tiny curvature dominated this test and rejected correct factorisations; the exact
                    #  count, inertia="ldl", accepts those)
                    coupled = (np.diff(A.tocsc().indptr) > 0) | (np.diff(W.tocsr().indptr) > 0)
                    dv = sol[:nv] * coupled
                    curv = dv @ (Hreg @ dv)
                    if curv >= 1e-11 * (dv @ dv) or (dv @ dv) == 0.0:
                        break
                    # The curvature of the computed step is a PROXY for IPOPT's test (the exact inertia MUMPS reports): the Newton step
                    # has a component in the range space of A', so d'Hd < 0 can happen with a positive definite reduced Hessian - the
                    # proxy then rejects a factorisation IPOPT accepts and delta_w is escalated for nothing (round 6: member 15 803 of the
                    # timed batch at iteration 6 - exact count 7 210 = m at delta_w = 1.37e-7, proxy: rejected, next accepted value 4.5e-3).
                    # inertia = "curvature_then_ldl": a rejection is checked with the exact count (dense Bunch-Kaufman LDL': 20 - 45 s
                    # for the 15 310 rows of industrial_poly - or 0.1 s with the sparse count without pivoting, _n_negative_sparse; on request only: the
                    # iterates of every stored oracle solve were produced with the proxy - the GPU parity test re-solves the members whose
                    # iteration count differs from the product's with it)
                    if o["inertia"] == "curvature_then_ldl":
                        nneg = _n_negative_sparse(Hreg, A, delta_c)
                        if nneg < 0:
                            nneg = _n_negative(sps.bmat([[Hreg, A.T], [A, -delta_c * sps.identity(m) if delta_c > 0 else None]], format="csc"))
                        stats["n_ldl_checks"] += 1
                        if nneg == m:
                            stats["n_proxy_false_rejections"] += 1
                            break
                stats["n_reg"] += 1
                if delta_w == 0.0:
                    delta_w = o["delta_w_0"] if delta_w_last == 0.0 else max(o["delta_w_min"], o["kappa_w_minus"] * delta_w_last)
                else:
                    delta_w *= o["kappa_w_plus_bar"] if (delta_w_last == 0.0 and first_try) else o["kappa_w_plus"]
                    if delta_w > o["delta_w_max"]:
                        if in_wd:       # (a tentative watchdog step led here: the watchdog ends like after an unacceptable last step)
                            break
                        raise RuntimeError("inertia correction failed")
                first_try = False
            if in_wd and delta_w > o["delta_w_max"]:
                wd_restart, in_wd, wd_count = wd["state"], False, 0
                continue
            if delta_w > 0:
                delta_w_last = delta_w
            dv, dy = sol[:nv], sol[nv:]
            dzl = np.where(has_l, mu / dl - zl - zl / dl * dv, 0.0)
            dzu = np.where(has_u, mu / du - zu + zu / du * dv, 0.0)

            def ftb(val, d):
                neg = d < 0
                if not neg.any():
                    return 1.0
                return min(1.0, np.min(-tau * val[neg] / d[neg]))

            a_max = min(ftb(dl[has_l], dv[has_l]), ftb(du[has_u], -dv[has_u]))
            a_z = min(ftb(zl[has_l], dzl[has_l]), ftb(zu[has_u], dzu[has_u]))

        # ---- filter line search
        theta = np.abs(c).sum()
        phi = barrier(fval, v)
        gphi = gf - np.where(has_l, mu / dl, 0.0) + np.where(has_u, mu / du, 0.0) + damp(mu)
        dphi = gphi @ dv
        if dphi < 0 and theta <= theta_min:
            a_min = o["gamma_alpha"] * min(o["gamma_theta"], o["gamma_phi"] * theta / (-dphi) if theta > 0 else np.inf,
                                           o["delta"] * theta ** o["s_theta"] / (-dphi) ** o["s_phi"] if theta > 0 else np.inf)
            if theta == 0:
                a_min = o["gamma_alpha"] * o["gamma_theta"]
        elif dphi < 0:
            a_min = o["gamma_alpha"] * min(o["gamma_theta"], o["gamma_phi"] * theta / (-dphi))
        else:
            a_min = o["gamma_alpha"] * o["gamma_theta"]
        a_min = max(a_min, 1e-14)

        def acceptable(th_t, ph_t, alpha, ref=None):
            theta_, phi_, dphi_ = ref if ref is not None else (theta, phi, dphi)      # (watchdog: the point where it started)
            if th_t > theta_max:
                return False, False
            for (tf, pf) in filt:
                if th_t >= tf and ph_t >= pf:
                    return False, False
            switching = dphi_ < 0 and alpha * (-dphi_) ** o["s_phi"] > o["delta"] * theta_ ** o["s_theta"]
            if theta_ <= theta_min and switching:
                eps_m = 10 * np.finfo(float).eps * abs(phi_)
                return ph_t - phi_ - eps_m <= o["eta_phi"] * alpha * dphi_, True
            eps_m = 10 * np.finfo(float).eps * abs(phi_)
            return (th_t <= (1 - o["gamma_theta"]) * theta_) or (ph_t - phi_ - eps_m <= -o["gamma_phi"] * theta_), False

        accepted = False
        n_ls = 0
        used_dv = dv
        augment_ref = None                  # filter augmentation w.r.t. this reference point (None: the current iterate)
        wd_done = False                     # this iteration's step was decided by the watchdog
        trig = o["watchdog_shortened_iter_trigger"]
        if trig > 0 and not in_wd and not skip_first and wd_count >= trig:
            in_wd, wd_iter = True, 0
            stats["n_watchdog"] += 1
            wd = dict(state=(v, y, zl, zu, fval, gval, c, A, gf, mu, tau, list(filt), delta_w, delta_w_last, lu, rx, dv, dy, dzl, dzu, a_max, a_z, dl, du),
                      ref=(theta, phi, dphi), alpha_test=a_max)
        if in_wd:
            alpha = a_max
            v_t = v + alpha * dv
            f_t, g_t = eval_fg(v_t[:n])
            c_t = cons(g_t, v_t[n:])
            th_t = np.abs(c_t).sum()
            ph_t = barrier(f_t, v_t) if np.isfinite(f_t) else np.inf
            ok_, armijo = acceptable(th_t, ph_t, wd["alpha_test"], wd["ref"]) if np.isfinite(ph_t) and np.isfinite(th_t) else (False, False)
            if ok_:
                accepted, wd_done, in_wd, wd_count = True, True, False, 0
                augment_ref = wd["ref"] + (wd["alpha_test"],)
            else:
                wd_iter += 1
                if wd_iter > o["watchdog_trial_iter_max"] or not (np.isfinite(ph_t) and np.isfinite(th_t)):
                    wd_restart, in_wd, wd_count = wd["state"], False, 0
                    continue
                accepted, wd_done, armijo = True, True, True      # taken without asking the filter; no filter entry
                augment_ref = False
        alpha = a_max * (0.5 if skip_first else 1.0) if not wd_done else alpha
        n_ls = 1 if (skip_first and not wd_done) else 0
        while alpha >= a_min and not wd_done:
            v_t = v + alpha * dv
            f_t, g_t = eval_fg(v_t[:n])
            c_t = cons(g_t, v_t[n:])
            th_t = np.abs(c_t).sum()
            ph_t = barrier(f_t, v_t) if np.isfinite(f_t) else np.inf
            ok_, armijo = acceptable(th_t, ph_t, alpha) if np.isfinite(ph_t) and np.isfinite(th_t) else (False, False)
            if ok_:
                accepted = True
                break
            # second-order correction on the first trial point
            if n_ls == 0 and th_t >= theta and o["max_soc"] > 0:
                c_soc = alpha * c + c_t
                th_old = theta
                for _ in range(o["max_soc"]):
                    stats["n_soc"] += 1
                    sol_s = lu.solve(-np.concatenate([rx, c_soc]))
                    dv_s = sol_s[:nv]
                    a_s = min(ftb(dl[has_l], dv_s[has_l]), ftb(du[has_u], -dv_s[has_u]))
                    v_s = v + a_s * dv_s
                    f_s, g_s = eval_fg(v_s[:n])
                    c_s = cons(g_s, v_s[n:])
                    th_s = np.abs(c_s).sum()
                    ph_s = barrier(f_s, v_s) if np.isfinite(f_s) else np.inf
                    ok_s, arm_s = acceptable(th_s, ph_s, a_s) if np.isfinite(ph_s) and np.isfinite(th_s) else (False, False)
                    if ok_s:
                        # the corrected direction replaces the Newton direction in every component (IPOPT:
                        # IpBacktrackingLineSearch uses actual_delta = delta_soc for the primal AND the dual step)
                        accepted, armijo = True, arm_s
                        v_t, f_t, g_t, c_t, th_t, ph_t = v_s, f_s, g_s, c_s, th_s, ph_s
                        dy = sol_s[nv:]
                        alpha_y = a_s
                        used_dv = dv_s
                        dzl = np.where(has_l, mu / dl - zl - zl / dl * dv_s, 0.0)
                        dzu = np.where(has_u, mu / du - zu + zu / du * dv_s, 0.0)
                        a_z = min(ftb(zl[has_l], dzl[has_l]), ftb(zu[has_u], dzu[has_u]))
                        break
                    if th_s > o["kappa_soc"] * th_old:
                        break
                    th_old = th_s
                    c_soc = a_s * c_soc + c_s
                if accepted:
                    alpha = alpha_y
                    break
            alpha *= 0.5
            n_ls += 1
        if not accepted:
            stats["n_ls_fail"] += 1
            alpha = max(alpha, a_min)
            v_t = v + alpha * dv
            f_t, g_t = eval_fg(v_t[:n])
            c_t = cons(g_t, v_t[n:])
            filt = []
            armijo = True
        if not wd_done:
            wd_count = wd_count + 1 if n_ls > 0 else 0          # consecutive iterations with a shortened step
        # ---- filter augmentation
        if augment_ref is False:
            pass
        elif augment_ref is not None:
            th_r, ph_r, dph_r, al_r = augment_ref
            switching = dph_r < 0 and al_r * (-dph_r) ** o["s_phi"] > o["delta"] * th_r ** o["s_theta"]
            if not (armijo and th_r <= theta_min and switching):
                filt.append(((1 - o["gamma_theta"]) * th_r, ph_r - o["gamma_phi"] * th_r))
        elif accepted and not armijo:
            filt.append(((1 - o["gamma_theta"]) * theta, phi - o["gamma_phi"] * theta))
        elif accepted and armijo:
            switching = dphi < 0 and alpha * (-dphi) ** o["s_phi"] > o["delta"] * theta ** o["s_theta"]
            if not (theta <= theta_min and switching):
                filt.append(((1 - o["gamma_theta"]) * theta, phi - o["gamma_phi"] * theta))
        # ---- accept
        v = v_t
        y = y + alpha * dy
        zl = zl + a_z * dzl
        zu = zu + a_z * dzu
        dl = np.where(has_l, v - vl, 1.0)
        du = np.where(has_u, vu - v, 1.0)
        ks = o["kappa_sigma"]
        zl = np.where(has_l, np.maximum(np.minimum(zl, ks * mu / dl), mu / (ks * dl)), 0.0)
        zu = np.where(has_u, np.maximum(np.minimum(zu, ks * mu / du), mu / (ks * du)), 0.0)
        fval, gval, c = f_t, g_t, c_t
        A = jac_full(v[:n])
        gf = grad_full(v[:n])
        stats["iters"].append(dict(alpha=alpha, alpha_z=a_z, mu=mu, delta_w=delta_w, n_ls=n_ls))
        it += 1

    xs = v[:n]
    res = Result()
    res["x"] = xs.copy()
    res["f"] = fval / sf
    res["g"] = gval / sg
    res["lam_g"] = lam_unscaled(y)
    res["lam_x"] = (zu[:n] - zl[:n]) / sf
    res["stats"] = dict(success=success, return_status=status, iter_count=it,
                        t_wall_total=time.perf_counter() - t_start, n_eval=n_eval,
                        n_ls_fail=stats["n_ls_fail"], n_reg=stats["n_reg"], n_soc=stats["n_soc"], n_watchdog=stats["n_watchdog"], mu=mu,
                        obj_scaling=sf, iters=stats["iters"], n_ldl_checks=stats["n_ldl_checks"],
                        n_proxy_false_rejections=stats["n_proxy_false_rejections"])
    return res
