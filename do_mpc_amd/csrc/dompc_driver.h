// dompc_driver.h - structured interior-point solver, part of dompc_kernel.h (included there, inside namespace dompc, in this order:
// dompc_edge.h, dompc_factor.h, dompc_node.h, dompc_riccati.h, dompc_forward.h, dompc_sweep.h, dompc_phases.h, dompc_driver.h).
// Contents: shared slack variables (nl_cons_single_slack); the interior-point driver solve_problem(); the bodies of the kernels.
// Sizes, record layouts, the thread context `Thr`, reductions and the small dense products are in dompc_kernel.h.

// ================================================================================================
// Shared slack variables (nl_cons_single_slack, EPS_GLOBAL).  The slack entries e (n_v = n_opt_x - off_eps of them, e_j is read by the
// nl_cons rows I_j of every edge whose parent node carries node_eps_off = off_eps + j - q) border the structured primal-dual system
//     [ K   B ] [ d  ]   [ -r   ]        K: the tree-structured system (x, u, w, s, lambda) the sweep + Riccati passes factorise,
//     [ B'  D ] [ de ] = [ -r_e ]        B = [0; E] with E = d c / d e (-1 in the rows I_j), D = Sigma_e + delta_w,
// r_e = grad_e f + E' lambda + barrier gradient.  The rows are LINEAR in e and B has entries in constraint rows only, so a structured
// solve with the constraint residual as an INPUT (the mode of the second-order correction, Prob::soc bit 0) delivers every product that is
// needed:  d(c + E v) - d(c) = -K^-1 [0; E] v  exactly.  Per iteration: the structured step d(c), one solve per slack for the columns
// of the Schur complement  S = D + E' (dlam(c + E_j) - dlam(c))_j  (symmetric positive definite iff the inertia of the bordered matrix is
// the right one: a failed Cholesky factorisation of S escalates delta_w like a failed factorisation inside the Riccati pass), the slack
// step  S de = -r_e - E' dlam(c),  and the final structured solve at the residual c + E de, which IS the structured part of the full
// Newton direction - nothing is accumulated from differences.  (n_v + 1 extra linear solves per iteration: the option is a convenience of
// the reference for small problems, not a throughput path.)  Same NLP, same variables as the reference: the iterates are IPOPT's.
DOMPC_DEV inline int epsg_off(const KArgs& A) { return A.node_eps_off[0]; }          // (the root reads eps[0, 0]: first entry of the block)
DOMPC_DEV inline int epsg_n(const KArgs& A) { return A.n_opt_x - A.node_eps_off[0]; }
// objective gradient and dual residual of the shared slacks at the current iterate (after every sweep of an iterate)
DOMPC_DEV inline void epsg_grad(const Thr& T, const Prob& Q) {
  const KArgs& A = *Q.A;
  const int o = epsg_off(A), nv = epsg_n(A);
  for (int j = T.tid; j < nv; j += T.nt) {
    double g = 0.0, r = 0.0;
    for (int e = 0; e < A.n_edges; ++e) {
      const int q = j - (A.node_eps_off[A.edge_parent[e]] - o);
      if (q < 0 || q >= NSE) continue;
      g += Q.sf * DOMPC_EPS_PEN[q];                                   // (the slack cost is added once per edge, _mpc.py:1254)
      const double* yd = Q.lam + A.edge_row0[e] + NW + NX;
      for (int i = 0; i < NE; ++i)
        if (nl_slack(i) == q) r -= yd[i] * Q.sgn[e * NE1 + i];
    }
    Q.gf[o + j] = g;
    Q.rd[o + j] = g + r - Q.zl[o + j] + Q.zu[o + j];
  }
  T.sync();
}
// -(E' v)_j = sum of v over the rows that read slack j
DOMPC_DEV inline double epsg_rowsum(const Prob& Q, int j, const double* v, const double* v0) {
  const KArgs& A = *Q.A;
  const int o = epsg_off(A);
  double t = 0.0;
  for (int e = 0; e < A.n_edges; ++e) {
    const int q = j - (A.node_eps_off[A.edge_parent[e]] - o);
    if (q < 0 || q >= NSE) continue;
    const int r0 = A.edge_row0[e] + NW + NX;
    for (int i = 0; i < NE; ++i)
      if (nl_slack(i) == q) t += (v[r0 + i] - (v0 ? v0[r0 + i] : 0.0)) * Q.sgn[e * NE1 + i];
  }
  return t;
}
// Q.c = Q.ct + E v on the rows that read a shared slack (v == nullptr: unit vector j1; j1 < 0 and v == nullptr: Q.c = Q.ct there)
DOMPC_DEV inline void epsg_residual(const Thr& T, const Prob& Q, const double* v, int j1) {
  const KArgs& A = *Q.A;
  const int o = epsg_off(A);
  for (int e = T.tid; e < A.n_edges; e += T.nt) {
    const int jo = A.node_eps_off[A.edge_parent[e]] - o;
    const int r0 = A.edge_row0[e] + NW + NX;
    for (int i = 0; i < NE; ++i) {
      const int q = nl_slack(i);
      if (q < 0) continue;
      const double ve = v ? v[jo + q] : ((jo + q == j1) ? 1.0 : 0.0);
      Q.c[r0 + i] = Q.ct[r0 + i] - ve * Q.sgn[e * NE1 + i];
    }
  }
  T.sync();
}

// ================================================================================================
DOMPC_DEV inline void solve_problem(const Thr& T, const KArgs& A, int b, int slot) {
  const dompc_options& O = A.opt;
  Prob Q = make_prob(A, slot, A.p + (int64_t)b * A.n_opt_p);
  const double* x0 = A.x0 + (int64_t)b * A.n_opt_x;
  const int nX = A.n_opt_x, nSl = A.n_edges * NE;
  int status = 2, it = 0, n_reg = 0, n_ls_fail = 0, n_sweeps = 0, n_trials = 0, n_soc = 0;

  // ---- bounds (relaxed, bound_relax_factor), starting point pushed inside, z = 1
  double cnt[2] = {0.0, 0.0};
  // A variable that is in no constraint, no cost term and has no bound (the collocation slots of the initial node in
  // every continuous model: _mpc.py:1061-1078 leaves stage 0 unbounded) is a zero row and column of the reference's
  // primal-dual matrix: its linear solver reports a singular system at delta_w = 0 in EVERY iteration and IPOPT
  // regularises (delta_w from the wrong-inertia rule, IpPDPerturbationHandler: PerturbForSingularity).  The structured
  // factorisation here never sees those variables, so the first attempt of an iteration is declared failed instead -
  // same delta_w sequence, same iterates (batch_reactor / rotating-masses goldens: 1e-11 instead of 1e-6 / 2e-5).
  bool singular0;
  {
    double fr[1] = {0.0};
    for (int d = T.tid; d < A.n_dummy; d += T.nt) {
      const int g = A.dummy_idx[d];
      if (!(A.lbx[g] > -INFINITY) && !(A.ubx[g] < INFINITY)) fr[0] += 1.0;
    }
    const int ops[1] = {R_SUM};
    wg_reduce(T, fr, ops);
    singular0 = fr[0] > 0.0;
  }
  for (int g = T.tid; g < nX; g += T.nt) {
    double l = A.lbx[g], u = A.ubx[g];
    if (l > -INFINITY) l -= fmin(O.constr_viol_tol, O.bound_relax_factor * fmax(1.0, fabs(l)));
    if (u < INFINITY) u += fmin(O.constr_viol_tol, O.bound_relax_factor * fmax(1.0, fabs(u)));
    double xv = x0[g];
    const bool hl = l > -INFINITY, hu = u < INFINITY;
    double pl = hl ? O.bound_push * fmax(1.0, fabs(l)) : 0.0;
    double pu = hu ? O.bound_push * fmax(1.0, fabs(u)) : 0.0;
    if (hl && hu) { pl = fmin(pl, O.bound_frac * (u - l)); pu = fmin(pu, O.bound_frac * (u - l)); }
    if (hl) xv = fmax(xv, l + pl);
    if (hu) xv = fmin(xv, u - pu);
    Q.lb_own[g] = l; Q.ub_own[g] = u; Q.x[g] = xv;
    Q.zl[g] = hl ? 1.0 : 0.0; Q.zu[g] = hu ? 1.0 : 0.0;
    if (sh_cnt(A, mk_x(A, g))) cnt[0] += (hl ? 1.0 : 0.0) + (hu ? 1.0 : 0.0);
  }
  for (int r = T.tid; r < A.n_g; r += T.nt) Q.lam[r] = 0.0;
  T.sync();
  // Variables that appear in no constraint and no cost term (unused scenario slots of the reference's opt_x struct,
  // SURVEY.md App. A.7) are not determined by the NLP, only by the barrier terms of their bounds.  Under `singular0` they
  // stay in the problem like in the reference - their barrier terms enter the line search, the step-size rules and the
  // error measures, and the delta_w of every iteration keeps their steps finite (CSTR golden: a one-sided one wanders to
  // 5e4 over five steps).  Without that regularisation (discrete models) the barrier alone drives a one-sided one to
  // +-1e160 over a few warm-started solves: there they are taken out - no bounds, no multipliers, value = the caller's
  // x0 entry projected onto its box.
  for (int d = T.tid; d < A.n_dummy; d += T.nt) {
    const int g = A.dummy_idx[d];
    if (singular0) continue;
    if (sh_cnt(A, mk_x(A, g))) cnt[0] -= (Q.lb_own[g] > -INFINITY ? 1.0 : 0.0) + (Q.ub_own[g] < INFINITY ? 1.0 : 0.0);
    Q.x[g] = fmin(fmax(x0[g], A.lbx[g]), A.ubx[g]);           // the caller's value, projected onto its box
    Q.lb_own[g] = -INFINITY; Q.ub_own[g] = INFINITY; Q.zl[g] = 0.0; Q.zu[g] = 0.0;
  }
  T.sync();
  if (A.lb_sh) {
    // the shared copy: final values only (every problem of the launch writes the same bits; another problem may be reading them)
    for (int g = T.tid; g < nX; g += T.nt) { A.lb_sh[g] = Q.lb_own[g]; A.ub_sh[g] = Q.ub_own[g]; }
    T.sync();
  }
  // slacks of the nl_cons rows: s = d(x) pushed into [lbg,ubg].  `rescale`: second call, after the scaling factors of the rows are known
  // (below): rows, bounds (relaxed first, then scaled - like IPOPT's scaled NLP) and slacks in scaled units.
  auto init_slacks = [&](bool rescale) {
    for (int e = T.tid; e < A.n_edges; e += T.nt) {
      for (int i = 0; i < NE; ++i) { Q.s[e * NE1 + i] = 0.0; if (!rescale) Q.sgn[e * NE1 + i] = 1.0; }
      if (DENSE_EDGE) dae_edge_f(Q, e, Q.x, Q.s, Q.ct); else eval_edge_f(Q, e, Q.x, Q.s, Q.ct);
      for (int i = 0; i < NE; ++i) {
        const int row = A.edge_row0[e] + NW + NX + i, si = e * NE1 + i;
        double l = A.lbg[row], u = A.ubg[row];
        if (l > -INFINITY) l -= fmin(O.constr_viol_tol, O.bound_relax_factor * fmax(1.0, fabs(l)));
        if (u < INFINITY) u += fmin(O.constr_viol_tol, O.bound_relax_factor * fmax(1.0, fabs(u)));
        l *= Q.sgn[si]; u *= Q.sgn[si];
        const bool hl = l > -INFINITY, hu = u < INFINITY;
        double pl = hl ? O.bound_push * fmax(1.0, fabs(l)) : 0.0;
        double pu = hu ? O.bound_push * fmax(1.0, fabs(u)) : 0.0;
        if (hl && hu) { pl = fmin(pl, O.bound_frac * (u - l)); pu = fmin(pu, O.bound_frac * (u - l)); }
        double sv = Q.ct[row];       // = d - 0
        if (hl) sv = fmax(sv, l + pl);
        if (hu) sv = fmin(sv, u - pu);
        Q.s[si] = sv; Q.sl[si] = l; Q.su[si] = u;
        Q.zsl[si] = hl ? 1.0 : 0.0; Q.zsu[si] = hu ? 1.0 : 0.0;
        if (!rescale && sh_cnt(A, mk_e(A, e))) cnt[1] += (hl ? 1.0 : 0.0) + (hu ? 1.0 : 0.0);
      }
    }
    T.sync();
  };
  if (NE > 0) init_slacks(false);
  {
    const int ops[2] = {R_SUM, R_SUM};
    wg_reduce(T, cnt, ops);
  }
  const double n_bounds = cnt[0] + cnt[1];
  const double n_dual = (double)(A.n_g - DOMPC_XROW_MASKED) + n_bounds;      // (masked extra row slots are no constraints: dompc_kernel.h, DOMPC_XROW)

  // ---- objective scaling from the gradient at the (pushed) starting point
  double mu = O.mu_init;
  Q.sf = 1.0;
  long long c_sweep = 0, c_bwd = 0, c_fwd = 0, c_ls = 0, c_meas = 0, c_ftb = 0, c_acc = 0, c_t = 0; const long long c_start = prof_clock();
  unsigned g_sweep = 0, g_bwd = 0, g_fwd = 0, g_ls = 0, g_meas = 0, g_ftb = 0, g_acc = 0, g_t = 0; const unsigned g_start = T.gen;      // device-scope barriers per phase (profile builds)
  if (T.tid == 0) T.fset(6, abort_requested(A));      // (read by everybody at the top of the loop, barriers in between)
  // (singular0: every iteration is regularised and delta_w is known before its sweep - folded into the condensed blocks
  //  there, Prob::dsw, instead of W'W being formed on demand by the Riccati pass: that path costs as much as the pass)
  auto delta_after = [&](double last) { return last == 0.0 ? O.delta_w_0 : fmax(O.delta_w_min, O.kappa_w_minus * last); };
  // ---- first sweep: gradient-based objective scaling and, for models without nl_cons rows, the least-squares estimate of
  // the constraint multipliers at the starting point (IPOPT section 3.6, option constr_mult_init_max):
  //     [I A'; A 0] (w, y) = -(grad f - z_L + z_U, 0),   y discarded if |y|_inf is above the limit.
  // The same structured solve as a Newton step, on a system in which the Hessian block is the identity: lambda = 0 (no
  // constraint curvature), objective Hessians left out (Prob::soc bit 1), z = 0 (no Sigma), delta = dsw = 1; the residual
  // is an input and zero (bit 0); the barrier gradient -mu/(x-l) + mu/(u-x) is the wanted -z_L + z_U = -1 + 1 when every
  // finite bound is moved one unit away from the point and mu = 1.  (nl_cons rows: their slack variables would need the
  // same treatment; IPOPT discards the estimate on the CSTR and kite examples anyway.)  The sweep of that solve is the one
  // that delivers the gradient for the objective scaling, so the estimate costs two Riccati passes and no extra sweep.
  const bool ls_init = NE == 0 && O.constr_mult_init_max > 0.0;
  if (ls_init) {
    for (int g = T.tid; g < nX; g += T.nt) {
      if (Q.lb_own[g] > -INFINITY) Q.lb_own[g] = Q.x[g] - 1.0;
      if (Q.ub_own[g] < INFINITY) Q.ub_own[g] = Q.x[g] + 1.0;
      Q.zl[g] = 0.0; Q.zu[g] = 0.0;
    }
    for (int r = T.tid; r < A.n_g; r += T.nt) Q.c[r] = 0.0;
    T.sync();
  }
  auto first_sweep = [&]() {
    ++n_sweeps;
    const int rc = ls_init ? run_sweep(T, Q, b, slot, 1.0, 3, 1.0) : run_sweep(T, Q, b, slot, mu, 0, singular0 ? delta_after(0.0) : 0.0);
    if (EPS_GLOBAL) epsg_grad(T, Q);
    return rc;
  };
  // ---- shared slack variables (EPS_GLOBAL): Schur complement on top of the structured solve, see epsg_* above.
  // workspace Q.gsc: S / its Cholesky factor (n_v x n_v, leading dimension NVG_MAX), then [flag | rhs / step (NVG_MAX)]
  // columns of the Schur complement after the structured step of this iterate (Q.dlam = dlam(c), kept in Q.dlam_e); returns 1 = wrong inertia
  auto epsg_build = [&](double delta) -> int {
    const int o = epsg_off(A), nv = epsg_n(A);
    double* G = Q.gsc;
    for (int g = T.tid; g < A.n_g; g += T.nt) { Q.dlam_e[g] = Q.dlam[g]; Q.ct[g] = Q.c[g]; }
    T.sync();
    int rc = 0;
    for (int j = 0; j < nv && !rc; ++j) {
      epsg_residual(T, Q, nullptr, j);
      ++n_sweeps;
      rc = run_sweep(T, Q, b, slot, mu, 1, delta);
      if (!rc) rc = run_backward(T, Q, b, slot, mu, delta);
      if (!rc) {
        run_forward(T, Q, b, slot, mu, delta);
        for (int jp = T.tid; jp < nv; jp += T.nt) G[jp * NVG_MAX + j] = -epsg_rowsum(Q, jp, Q.dlam, Q.dlam_e);
      }
    }
    epsg_residual(T, Q, nullptr, -1);                    // Q.c back to c(x)
    if (T.tid == 0) {
      int ok = rc ? 0 : 1;
      for (int j = 0; j < nv && ok; ++j) {
        const int g = o + j;
        G[j * NVG_MAX + j] += sigma_of(Q.x[g], Q.lb[g], Q.ub[g], Q.zl[g], Q.zu[g]) + delta;
      }
      for (int j = 0; j < nv && ok; ++j) {               // Cholesky, lower triangle in place
        double dj = G[j * NVG_MAX + j];
        for (int k = 0; k < j; ++k) dj -= G[j * NVG_MAX + k] * G[j * NVG_MAX + k];
        if (!(dj > 0.0)) { ok = 0; break; }
        dj = sqrt(dj);
        G[j * NVG_MAX + j] = dj;
        for (int i = j + 1; i < nv; ++i) {
          double t = 0.5 * (G[i * NVG_MAX + j] + G[j * NVG_MAX + i]);      // (S is symmetric up to rounding)
          for (int k = 0; k < j; ++k) t -= G[i * NVG_MAX + k] * G[j * NVG_MAX + k];
          G[i * NVG_MAX + j] = t / dj;
        }
      }
      G[NVG_MAX * NVG_MAX] = ok ? 0.0 : 1.0;
    }
    T.sync();
    return G[NVG_MAX * NVG_MAX] != 0.0;
  };
  // slack step and the structured part of the full direction, given the structured step at the CURRENT residual Q.c (its
  // multiplier steps in `dl`) and the factor of S; Q.ct is free at both call sites (the trial values have been consumed)
  auto epsg_apply = [&](double delta, const double* dl) -> int {
    const int o = epsg_off(A), nv = epsg_n(A);
    double* G = Q.gsc;
    double* de = G + NVG_MAX * NVG_MAX + 1;
    for (int j = T.tid; j < nv; j += T.nt) {
      const int g = o + j;
      const double re = Q.rd[g] + Q.zl[g] - Q.zu[g] + bar_grad(Q.x[g], Q.lb[g], Q.ub[g], mu);
      de[j] = -re + epsg_rowsum(Q, j, dl, nullptr);       // -r_e - E' dlam(c)
    }
    for (int g = T.tid; g < A.n_g; g += T.nt) Q.ct[g] = Q.c[g];
    T.sync();
    if (T.tid == 0) {
      for (int i = 0; i < nv; ++i) {
        double t = de[i];
        for (int k = 0; k < i; ++k) t -= G[i * NVG_MAX + k] * de[k];
        de[i] = t / G[i * NVG_MAX + i];
      }
      for (int i = nv - 1; i >= 0; --i) {
        double t = de[i];
        for (int k = i + 1; k < nv; ++k) t -= G[k * NVG_MAX + i] * de[k];
        de[i] = t / G[i * NVG_MAX + i];
      }
    }
    T.sync();
    epsg_residual(T, Q, de, -1);
    ++n_sweeps;
    int rc = run_sweep(T, Q, b, slot, mu, 1, delta);
    if (!rc) rc = run_backward(T, Q, b, slot, mu, delta);
    if (!rc) run_forward(T, Q, b, slot, mu, delta);
    epsg_residual(T, Q, nullptr, -1);
    for (int j = T.tid; j < nv; j += T.nt) Q.dx[o + j] = de[j];
    T.sync();
    return rc;
  };
  int bad = first_sweep();
  // ---- IPOPT's gradient-based scaling of the CONSTRAINTS (nlp_scaling_method = gradient-based, same option as the objective scaling):
  // a row whose gradient at the starting point has a max-norm above nlp_scaling_max_gradient (100) is multiplied by 100 / that norm.
  // Restated for the nl_cons rows (kite example: the height constraint, gradient 335 - a soft row: sg (d(x, u) - eps) <= sg ub): through
  // the row's slack and its bound multipliers the factor changes the iterates from the first step on.  Rows of the dynamics: the Newton
  // step is invariant under their scaling and none of the examples has such a row above 100 apart from the dynamic bicycle (179, same
  // iterates as the oracle, which scales them) - not scaled here.
  if (NE > 0 && O.obj_scaling && !bad) {
    double any[1] = {0.0};
    for (int e = T.tid; e < A.n_edges; e += T.nt) {
      if (!mk_e(A, e)) continue;
      for (int i = 0; i < NE; ++i) {
        double gm = nl_slack(i) >= 0 ? 1.0 : 0.0;        // (the column of the row's slack variable `_eps`)
        for (int a = 0; a < NA; ++a) gm = fmax(gm, fabs(Q.EW(e, EW_JD + i * NA + a)));
        if (DENSE_EDGE) for (int c = 0; c < NW; ++c) gm = fmax(gm, fabs(Q.EW(e, EW_JDW + i * NW + c)));
        if (gm > O.nlp_scaling_max_gradient) { Q.sgn[e * NE1 + i] = fmax(O.nlp_scaling_max_gradient / gm, 1e-8); any[0] = 1.0; }
      }
    }
    const int ops[1] = {R_MAX};
    wg_reduce(T, any, ops);
    if (any[0] > 0.0) {
      T.sync();
      init_slacks(true);
      bad = first_sweep();
    }
  }
  if (O.obj_scaling) {
    double gm[1] = {0.0};
    for (int g = T.tid; g < nX; g += T.nt)
      if (sh_cnt(A, mk_x(A, g))) gm[0] = fmax(gm[0], fabs(Q.gf[g]));
    const int ops[1] = {R_MAX};
    wg_reduce(T, gm, ops);
    if (gm[0] > O.nlp_scaling_max_gradient) {
      Q.sf = fmax(O.nlp_scaling_max_gradient / gm[0], 1e-8);
      bad = first_sweep();
    }
  }
  if (ls_init) {
    int ls_bad = bad;
    if (!ls_bad) ls_bad = run_backward(T, Q, b, slot, 1.0, 1.0, 2);
    if (!ls_bad) run_forward(T, Q, b, slot, 1.0, 1.0);
    double ym[1] = {0.0};
    for (int r = T.tid; r < A.n_g; r += T.nt) {
      if (!mk_g(A, r)) continue;                        // (tree sharding: the rows this rank computes)
      const double y = Q.dlam[r];
      ym[0] = fmax(ym[0], (y == y) ? fabs(y) : INFINITY);
    }
    {
      const int ops[1] = {R_MAX};
      wg_reduce(T, ym, ops);
    }
    const bool keep = !ls_bad && ym[0] <= O.constr_mult_init_max;
    for (int r = T.tid; r < A.n_g; r += T.nt) Q.lam[r] = keep ? Q.dlam[r] : 0.0;
    for (int g = T.tid; g < nX; g += T.nt) {             // bounds and bound multipliers back to their starting values
      double l = A.lbx[g], u = A.ubx[g];
      const bool hl = Q.lb_own[g] > -INFINITY, hu = Q.ub_own[g] < INFINITY;   // (unused variables that were taken out stay out)
      if (hl) Q.lb_own[g] = l - fmin(O.constr_viol_tol, O.bound_relax_factor * fmax(1.0, fabs(l)));
      if (hu) Q.ub_own[g] = u + fmin(O.constr_viol_tol, O.bound_relax_factor * fmax(1.0, fabs(u)));
      Q.zl[g] = hl ? 1.0 : 0.0; Q.zu[g] = hu ? 1.0 : 0.0;
    }
    T.sync();
    bad = run_sweep(T, Q, b, slot, mu, 0, singular0 ? delta_after(0.0) : 0.0);
    ++n_sweeps;
  }
  const double mu_min = fmin(O.tol, O.compl_inf_tol * Q.sf) / (O.kappa_eps + 1.0);
  double tau = fmax(O.tau_min, 1.0 - mu);
  if (KAPPA_D != 0.0) Q.mu = mu;              // (read by measure() for the damping term of the dual residual)
  Errs E = measure(T, Q, nullptr);
  const double theta0 = E.theta;
  const double theta_max = 1e4 * fmax(1.0, theta0), theta_min = 1e-4 * fmax(1.0, theta0);
  // barrier sum -sum log(x - l) - sum log(u - x) of the starting point; afterwards it is carried over from the line search
  double bar_sum;
  {
    double bs[1] = {0.0};
    LogAcc La{1.0, 0, 0};
    double lin = 0.0;
    double x_[DOMPC_FW], l_[DOMPC_FW], u2_[DOMPC_FW];
#define L_(u, g) x_[u] = Q.x[g]; l_[u] = Q.lb[g]; u2_[u] = Q.ub[g];
#define B_(u, g)                                                       \
    if (sh_cnt(A, mk_x(A, g))) {                                       \
      if (l_[u] > -INFINITY) logacc_add(La, x_[u] - l_[u]);            \
      if (u2_[u] < INFINITY) logacc_add(La, u2_[u] - x_[u]);           \
      if (KAPPA_D != 0.0) { const double os_ = one_sided(l_[u], u2_[u]); lin += os_ > 0.0 ? x_[u] - l_[u] : (os_ < 0.0 ? u2_[u] - x_[u] : 0.0); } \
    }
    DOMPC_FOR4(nX, L_, B_)
#undef L_
#undef B_
    for (int g = T.tid; g < nSl; g += T.nt) {
      if (!sh_cnt(A, mk_e(A, g / NE1))) continue;
      const int si = (g / NE1) * NE1 + g % NE1;
      if (Q.sl[si] > -INFINITY) logacc_add(La, Q.s[si] - Q.sl[si]);
      if (Q.su[si] < INFINITY) logacc_add(La, Q.su[si] - Q.s[si]);
      if (KAPPA_D != 0.0) { const double os_ = one_sided(Q.sl[si], Q.su[si]); lin += os_ > 0.0 ? Q.s[si] - Q.sl[si] : (os_ < 0.0 ? Q.su[si] - Q.s[si] : 0.0); }
    }
    bs[0] = -logacc_value(La);
    if (KAPPA_D != 0.0) bs[0] += KAPPA_D * lin;
    const int ops[1] = {R_SUM};
    wg_reduce(T, bs, ops);
    bar_sum = bs[0];
  }
  int n_filt = 0;
  double delta_last = 0.0;
  int acc_count = 0;
  const double s_max = 100.0;
  double E0 = 0.0;
  // IPOPT's watchdog procedure (IpBacktrackingLineSearch; options watchdog_shortened_iter_trigger = 10, watchdog_trial_iter_max = 3; not
  // in the 2006 paper): after `trigger` consecutive iterations whose step was shortened by the backtracking, up to `trial_iter_max` full
  // fraction-to-the-boundary steps are taken without asking the filter, each tested against the point where the watchdog STARTED
  // (its theta, barrier objective, directional derivative and step size); none acceptable: back to that point and the direction
  // computed there, regular backtracking from the second trial step size.  It is what keeps non-convex problems from crawling with
  // 2^-10 steps for hundreds of iterations (kite over the full horizon: 87 instead of 906 iterations, the oracle's 87 with exact inertia,
  // profiles/r04_crawl_traces.txt).  State: the iterate in Q.*_wd, its direction in Q.dx_sv / dlam_sv / ds_sv (no second-order
  // correction runs while a watchdog is active), scalars below.
  int wd_count = 0, wd_iter = 0, n_watchdog = 0;
  bool in_wd = false, wd_resume = false;
  double wd_theta = 0.0, wd_phi = 0.0, wd_dphi = 0.0, wd_alpha = 0.0, wd_amax = 0.0, wd_az = 0.0, wd_delta = 0.0, wd_delta_last = 0.0, wd_bar = 0.0;
  Errs wd_E = E;
  double delta = 0.0, a_max = 1.0, a_z = 1.0, dphi = 0.0;

  while (true) {
    bool skip_first = false;
    if (wd_resume) {
      // the watchdog gave up: back at the point where it started, with the direction computed there
      for (int g = T.tid; g < nX; g += T.nt) { Q.x[g] = Q.x_wd[g]; Q.zl[g] = Q.zl_wd[g]; Q.zu[g] = Q.zu_wd[g]; Q.dx[g] = Q.dx_sv[g]; }
      for (int g = T.tid; g < A.n_g; g += T.nt) { Q.lam[g] = Q.lam_wd[g]; Q.dlam[g] = Q.dlam_sv[g]; }
      for (int g = T.tid; g < nSl; g += T.nt) { Q.s[g] = Q.s_wd[g]; Q.zsl[g] = Q.zsl_wd[g]; Q.zsu[g] = Q.zsu_wd[g]; Q.ds[g] = Q.ds_sv[g]; }
      T.sync();
      E = wd_E; bar_sum = wd_bar; delta = wd_delta; delta_last = wd_delta_last; a_max = wd_amax; a_z = wd_az; dphi = wd_dphi;
      wd_resume = false;
      skip_first = true;
    } else {
    // (a tentative watchdog step that leads to a point where the step computation fails - sweep, inertia correction, NaN - ends the
    //  watchdog like an unacceptable third step: back to the stored point)
    if (bad) { if (in_wd) { bad = 0; wd_resume = true; in_wd = false; wd_count = 0; continue; } status = 3; break; }
    if (T.fget(6)) { status = 6; break; }                                    // the host asked the kernel to stop
    if (((WIDE_OK && T.nwg > 1) || sh_on(A)) && T.fget(7)) { status = 5; break; }       // a peer workgroup never arrived at a barrier
    const double sd = fmax(s_max, (E.sum_y + E.C.sum_z) / fmax(1.0, n_dual)) / s_max;
    const double sc = fmax(s_max, E.C.sum_z / fmax(1.0, n_bounds)) / s_max;
    const double e_c0 = comp_err(E.C, 0.0);
    E0 = fmax(E.e_d / sd, fmax(E.e_p, e_c0 / sc));
    if (!(E0 == E0) || !(E.obj == E.obj)) { if (in_wd) { wd_resume = true; in_wd = false; wd_count = 0; continue; } status = 4; break; }
    if (E0 <= O.tol && E.e_d <= O.dual_inf_tol && E.e_p <= O.constr_viol_tol && e_c0 <= O.compl_inf_tol) {
      status = 0; break;
    }
    if (E0 <= O.acceptable_tol) {
      if (++acc_count >= O.acceptable_iter) { status = 1; break; }
    } else acc_count = 0;
    if (it >= O.max_iter) { status = 2; break; }

    // ---- barrier update (monotone Fiacco-McCormick)
    const double mu_before = mu;
    while (true) {       // (the iterate does not move in here: only the complementarity error depends on mu - comp_err)
      const double Emu = fmax(E.e_d / sd, fmax(E.e_p, comp_err(E.C, mu) / sc));
      if (Emu <= O.kappa_eps * mu && mu > mu_min) {
        mu = fmax(mu_min, fmin(O.kappa_mu * mu, pow(mu, O.theta_mu)));
        tau = fmax(O.tau_min, 1.0 - mu);
        n_filt = 0;
        in_wd = false; wd_count = 0;      // (a new barrier problem: the watchdog's reference point is void)
      } else break;
    }
    if (mu != mu_before) {
      if (QUAD_FWD && forward_adjoint(Q, mu) && !Q.lu_ok) {
        // the barrier parameter fell by more than one level at this iterate, into the range of the adjoint variant of the forward pass, and
        // the last sweep did not store the inverses that variant reads (lu_store_rule): the sweep is repeated at the new mu, storing them
        ++n_sweeps;
        if (run_sweep(T, Q, b, slot, mu, 4, Q.dsw)) { if (in_wd) { wd_resume = true; in_wd = false; wd_count = 0; continue; } status = 3; break; }
      } else {
        refresh_mu(T, Q, mu - mu_before);
        Q.rp_mu = mu;
      }
    }

    // ---- search direction with inertia correction (delta_w on all primal variables)
    delta = 0.0;
    bool first_try = true, dir_ok = true, recs_dirty = false;
    while (true) {
      c_t = prof_clock(); g_t = T.gen;
      int fail = (singular0 && delta == 0.0) ? 1 : run_backward(T, Q, b, slot, mu, delta);
      if (EPS_GLOBAL && !fail) {
        run_forward(T, Q, b, slot, mu, delta);            // structured step at c(x), then the Schur complement of the shared slacks
        fail = epsg_build(delta);
        recs_dirty = true;                                // (the vector parts of the records now belong to the last column's residual)
      }
      c_bwd += prof_clock() - c_t; g_bwd += T.gen - g_t;
      if (!fail) break;
      if (delta == 0.0) {
        delta = delta_after(delta_last);
      } else {
        delta *= (delta_last == 0.0 && first_try) ? O.kappa_w_plus_bar : O.kappa_w_plus;
        first_try = false;   // (IPOPT: the larger factor only on the very first increase)
        if (delta > O.delta_w_max) { dir_ok = false; break; }
      }
      if ((NW > 0 && delta != Q.dsw) || (EPS_GLOBAL && recs_dirty)) {
        // the condensed blocks hold another inertia correction (Q~(delta) = Q~ + delta W'W, q~ likewise): the sweep is
        // repeated with this one folded in - W is not kept beyond the sweep, so the Riccati pass cannot add the
        // difference itself.  Rare: under `singular0` the first delta of an iteration is known before its sweep.
        ++n_sweeps;
        if (run_sweep(T, Q, b, slot, mu, 0, delta)) { dir_ok = false; break; }
        recs_dirty = false;
      }
    }
    if (!dir_ok) { if (in_wd) { wd_resume = true; in_wd = false; wd_count = 0; continue; } status = 3; break; }
    if (delta > 0.0) { delta_last = delta; ++n_reg; }
    c_t = prof_clock(); g_t = T.gen;
    if (EPS_GLOBAL) { if (epsg_apply(delta, Q.dlam_e)) { if (in_wd) { wd_resume = true; in_wd = false; wd_count = 0; continue; } status = 3; break; } }
    else run_forward(T, Q, b, slot, mu, delta);
    c_fwd += prof_clock() - c_t; g_fwd += T.gen - g_t;

    // ---- fraction to the boundary, directional derivative of the barrier function
    c_t = prof_clock(); g_t = T.gen;
    // largest ratios (-dx)/(x - l), dx/(u - x) and (-dz)/z over the bounded variables: the fraction-to-the-boundary steps
    // are tau / ratio (one division at the end instead of one per bound), and the directional derivative of the barrier function
    double r5[5];
    run_step_rules(T, Q, b, slot, mu, r5);
    a_max = (r5[0] > tau) ? tau / r5[0] : 1.0; dphi = r5[2];
    a_z = (r5[1] > tau) ? tau / r5[1] : 1.0;
    c_ftb += prof_clock() - c_t; g_ftb += T.gen - g_t;
    }     // (!wd_resume)
    auto step_rules = [&](double (&r5)[5]) { run_step_rules(T, Q, b, slot, mu, r5); };
    const double theta = E.theta;
    const double phi = E.obj + mu * bar_sum;       // (the barrier sum of the current point was formed when it was a trial point)

    c_t = prof_clock(); g_t = T.gen;
    // ---- filter line search with second-order correction (no restoration phase)
    const double gamma_theta = 1e-5, gamma_phi = 1e-8, eta_phi = 1e-8, s_theta = 1.1, s_phi = 2.3, gamma_alpha = 0.05;
    const double kappa_soc = 0.99;
    double a_min;
    if (dphi < 0.0 && theta <= theta_min)
      a_min = (theta > 0.0) ? gamma_alpha * fmin(gamma_theta, fmin(gamma_phi * theta / (-dphi),
                                                                    pow(theta, s_theta) / pow(-dphi, s_phi)))
                            : gamma_alpha * gamma_theta;
    else if (dphi < 0.0) a_min = gamma_alpha * fmin(gamma_theta, gamma_phi * theta / (-dphi));
    else a_min = gamma_alpha * gamma_theta;
    a_min = fmax(a_min, 1e-14);
    // objective, constraint violation and barrier sum of the trial point x + al * dx (left in Q.xt / Q.st, constraint values in Q.ct)
    auto eval_trial = [&](double al, double& obj_o, double& th_o, double& bar_o) {
      run_eval_trial(T, Q, b, slot, al, obj_o, th_o, bar_o);
      ++n_trials;
    };
    // filter / sufficient-decrease tests of a trial point reached with step size al (IPOPT eqs. (18)-(20))
    auto acceptable_ref = [&](double th_, double ph_, double al, bool& armijo_case, double theta, double phi, double dphi) -> bool {
      armijo_case = false;
      bool ok = (ph_ == ph_) && (th_ == th_) && fabs(ph_) < INFINITY && th_ <= theta_max;
      if (ok) {
        for (int q = 0; q < n_filt; ++q)
          if (th_ >= T.filt[2 * q] && ph_ >= T.filt[2 * q + 1]) { ok = false; break; }
      }
      if (ok) {
        const bool switching = dphi < 0.0 && al * pow(-dphi, s_phi) > pow(theta, s_theta);
        const double eps_m = 10.0 * 2.220446049250313e-16 * fabs(phi);
        if (theta <= theta_min && switching) {
          armijo_case = true;
          ok = (ph_ - phi - eps_m <= eta_phi * al * dphi);
        } else {
          ok = (th_ <= (1.0 - gamma_theta) * theta) || (ph_ - phi - eps_m <= -gamma_phi * theta);
        }
      }
      return ok;
    };
    auto acceptable = [&](double th_, double ph_, double al, bool& armijo_case) -> bool { return acceptable_ref(th_, ph_, al, armijo_case, theta, phi, dphi); };
    // corrected constraint residual of the second-order correction: c <- al * c + c(trial point)   (IPOPT eq. (27))
    auto soc_residual = [&](double al) {
      double c_[DOMPC_FW], ct_[DOMPC_FW];
#define L_(u, g) c_[u] = Q.c[g]; ct_[u] = Q.ct[g];
#define B_(u, g) if (mk_g(A, g)) Q.c[g] = al * c_[u] + ct_[u];
      DOMPC_FOR4(A.n_g, L_, B_)
#undef L_
#undef B_
      T.sync();
    };
    // the Newton direction is set aside while corrected directions are tried (nothing else of the regular solve is needed
    // again: the per-edge records and Q.c are rebuilt by the sweep of the next iterate)
    auto keep_direction = [&](bool restore) {
      for (int g = T.tid; g < nX; g += T.nt) { if (restore) Q.dx[g] = Q.dx_sv[g]; else Q.dx_sv[g] = Q.dx[g]; }
      for (int g = T.tid; g < A.n_g; g += T.nt) { if (restore) Q.dlam[g] = Q.dlam_sv[g]; else Q.dlam_sv[g] = Q.dlam[g]; }
      for (int g = T.tid; g < nSl; g += T.nt) { if (restore) Q.ds[g] = Q.ds_sv[g]; else Q.ds_sv[g] = Q.ds[g]; }
      T.sync();
    };
    double alpha = skip_first ? 0.5 * a_max : a_max;
    bool accepted = false, armijo_used = false, stale = false;
    double th_t = 0.0, obj_t = 0.0, bar_t = bar_sum;
    int n_ls = skip_first ? 1 : 0;
    bool wd_done = false, wd_augment_ref = false, wd_no_augment = false;
    if (O.watchdog_shortened_iter_trigger > 0 && !in_wd && !skip_first && wd_count >= O.watchdog_shortened_iter_trigger) {
      in_wd = true; wd_iter = 0; ++n_watchdog;
      for (int g = T.tid; g < nX; g += T.nt) { Q.x_wd[g] = Q.x[g]; Q.zl_wd[g] = Q.zl[g]; Q.zu_wd[g] = Q.zu[g]; Q.dx_sv[g] = Q.dx[g]; }
      for (int g = T.tid; g < A.n_g; g += T.nt) { Q.lam_wd[g] = Q.lam[g]; Q.dlam_sv[g] = Q.dlam[g]; }
      for (int g = T.tid; g < nSl; g += T.nt) { Q.s_wd[g] = Q.s[g]; Q.zsl_wd[g] = Q.zsl[g]; Q.zsu_wd[g] = Q.zsu[g]; Q.ds_sv[g] = Q.ds[g]; }
      T.sync();
      wd_E = E; wd_bar = bar_sum; wd_delta = delta; wd_delta_last = delta_last; wd_amax = a_max; wd_az = a_z;
      wd_theta = theta; wd_phi = phi; wd_dphi = dphi; wd_alpha = a_max;
    }
    if (in_wd) {
      eval_trial(alpha, obj_t, th_t, bar_t);
      bool armijo_case = false;
      if (acceptable_ref(th_t, obj_t + mu * bar_t, wd_alpha, armijo_case, wd_theta, wd_phi, wd_dphi)) {
        accepted = true; armijo_used = armijo_case; wd_done = true; wd_augment_ref = true;
        in_wd = false; wd_count = 0;
      } else {
        ++wd_iter;
        const double ph_ = obj_t + mu * bar_t;
        if (wd_iter > O.watchdog_trial_iter_max || !(ph_ == ph_) || !(th_t == th_t) || !(fabs(ph_) < INFINITY)) {
          wd_resume = true; in_wd = false; wd_count = 0;
          continue;
        }
        accepted = true; wd_done = true; wd_no_augment = true;       // taken without asking the filter; no filter entry
      }
    }
    while (!wd_done) {
      eval_trial(alpha, obj_t, th_t, bar_t);
      stale = false;
      bool armijo_case = false;
      if (acceptable(th_t, obj_t + mu * bar_t, alpha, armijo_case)) { accepted = true; armijo_used = armijo_case; break; }
      if (n_ls == 0 && O.max_soc > 0 && th_t >= theta) {
        // Second-order correction (IPOPT section 2.4): the full step was rejected and did not reduce the constraint violation.
        // Solve the SAME linear system again with the corrected residual c_soc = alpha c(x) + c(x + alpha d): the sweep is repeated
        // with the residual as an input (all matrices come out identical; only the vector parts of the records change),
        // followed by the two Riccati passes.  Accepted: the corrected direction replaces the Newton direction (step size,
        // multiplier steps and all).  Not accepted: the Newton direction comes back from its copy.  First trial of an
        // iteration only; on the industrial_poly benchmark 1.2 corrections per cold solve (57 iterations).
        double th_old = theta;
        keep_direction(false);
        soc_residual(alpha);
        bool soc_ok = false;
        for (int k = 0; k < O.max_soc; ++k) {
          ++n_soc; ++n_sweeps;
          if (run_sweep(T, Q, b, slot, mu, 1, delta)) break;
          if (run_backward(T, Q, b, slot, mu, delta)) break;
          run_forward(T, Q, b, slot, mu, delta);
          if (EPS_GLOBAL && epsg_apply(delta, Q.dlam)) break;
          double q5[5];
          step_rules(q5);
          const double a_s = (q5[0] > tau) ? tau / q5[0] : 1.0;
          double obj_s = 0.0, th_s = 0.0, bar_s = 0.0;
          eval_trial(a_s, obj_s, th_s, bar_s);
          bool arm_s = false;
          if (acceptable(th_s, obj_s + mu * bar_s, a_s, arm_s)) {
            accepted = true; armijo_used = arm_s; soc_ok = true;
            alpha = a_s;
            a_z = (q5[1] > tau) ? tau / q5[1] : 1.0;
            obj_t = obj_s; th_t = th_s; bar_t = bar_s;
            break;
          }
          if (!(th_s <= kappa_soc * th_old)) break;
          th_old = th_s;
          soc_residual(a_s);
        }
        if (soc_ok) break;
        keep_direction(true);                             // back to the Newton direction of this iterate
        stale = true;                                     // (Q.xt / Q.st / Q.ct hold the last corrected trial point)
      }
      if (!(alpha * 0.5 >= a_min)) break;  // xt/st/ct stay at the last evaluated alpha (also leaves on a NaN step size)
      alpha *= 0.5;
      ++n_ls;
    }
    if (bad) { if (in_wd) { bad = 0; wd_resume = true; in_wd = false; wd_count = 0; continue; } status = 3; break; }      // (same rule as at the top of the loop)
    if (!accepted && stale) eval_trial(alpha, obj_t, th_t, bar_t);
    if (!wd_done) wd_count = n_ls > 0 ? wd_count + 1 : 0;         // consecutive iterations with a shortened step
    if (!accepted) {
      // no restoration phase: take the smallest trial step and reset the filter
      ++n_ls_fail;
      n_filt = 0;
    } else if (!armijo_used && !wd_no_augment) {
      if (T.ltid == 0) {
        int q = n_filt < MAX_FILTER ? n_filt : MAX_FILTER - 1;
        T.filt[2 * q] = (1.0 - gamma_theta) * (wd_augment_ref ? wd_theta : theta);
        T.filt[2 * q + 1] = (wd_augment_ref ? wd_phi : phi) - gamma_phi * (wd_augment_ref ? wd_theta : theta);
      }
      if (n_filt < MAX_FILTER) ++n_filt;
      T.lsync();
    }
    bar_sum = bar_t;                  // xt of the last evaluated trial becomes the iterate
    c_ls += prof_clock() - c_t; g_ls += T.gen - g_t;
    // ---- accept the trial point
    c_t = prof_clock(); g_t = T.gen;
    const Comp Cp = run_accept(T, Q, b, slot, alpha, a_z, mu);
    if (A.trace && b == 0 && T.tid == 0 && it < A.trace_cap) {
      double* tr = A.trace + 8 * it;
      tr[0] = it; tr[1] = mu; tr[2] = E0; tr[3] = E.e_p; tr[4] = E.e_d; tr[5] = accepted ? alpha : -alpha;
      tr[6] = delta; tr[7] = E.obj / Q.sf;
    }
    if (T.tid == 0) T.fset(6, abort_requested(A));
    T.sync();
    c_acc += prof_clock() - c_t; g_acc += T.gen - g_t;
    ++it;
    c_t = prof_clock(); g_t = T.gen; bad = run_sweep(T, Q, b, slot, mu, 0, singular0 ? delta_after(delta_last) : 0.0); c_sweep += prof_clock() - c_t; g_sweep += T.gen - g_t;
    ++n_sweeps;
    if (EPS_GLOBAL) epsg_grad(T, Q);
    if (KAPPA_D != 0.0) Q.mu = mu;
    c_t = prof_clock(); g_t = T.gen; E = measure(T, Q, &Cp); c_meas += prof_clock() - c_t; g_meas += T.gen - g_t;
  }

  // ---- outputs (unscaled multipliers, CasADi sign convention)
  if (A.trace && b == 0 && T.tid == 0 && A.trace_cap > 8) { double* tr = A.trace + 8 * (A.trace_cap - 1); tr[0] = (double)c_sweep; tr[1] = (double)c_bwd; tr[2] = (double)c_fwd; tr[3] = (double)c_ls; tr[4] = (double)c_meas; tr[5] = (double)(prof_clock() - c_start); tr[6] = (double)c_ftb; tr[7] = (double)c_acc; if (T.prof) { double* t2 = A.trace + 8 * (A.trace_cap - 2); for (int i = 0; i < 8; ++i) t2[i] = (double)T.prof[i]; double* t3 = A.trace + 8 * (A.trace_cap - 3); for (int i = 0; i < 8; ++i) t3[i] = (double)T.prof[8 + i]; double* t4 = A.trace + 8 * (A.trace_cap - 4); for (int i = 0; i < 8; ++i) t4[i] = (double)T.prof[16 + i]; if (A.trace_cap > 12) { double* t5 = A.trace + 8 * (A.trace_cap - 5); for (int i = 0; i < 8; ++i) t5[i] = (double)T.prof[24 + i]; double* t6 = A.trace + 8 * (A.trace_cap - 6); t6[0] = g_sweep; t6[1] = g_bwd; t6[2] = g_fwd; t6[3] = g_ls; t6[4] = g_meas; t6[5] = T.gen - g_start; t6[6] = g_ftb; t6[7] = g_acc; } } }
  const double isf = 1.0 / Q.sf;
  // (sharded problem: every entry is written by exactly one rank, zeros elsewhere -> a SUM over the ranks is the full vector)
  if (A.x_out) for (int g = T.tid; g < nX; g += T.nt) A.x_out[(int64_t)b * nX + g] = sh_cnt(A, mk_x(A, g)) ? Q.x[g] : 0.0;
  if (A.lam_x_out) for (int g = T.tid; g < nX; g += T.nt) A.lam_x_out[(int64_t)b * nX + g] = sh_cnt(A, mk_x(A, g)) ? (Q.zu[g] - Q.zl[g]) * isf : 0.0;
  if (A.lam_g_out) for (int r = T.tid; r < A.n_g; r += T.nt) A.lam_g_out[(int64_t)b * A.n_g + r] = sh_cnt(A, mk_g(A, r)) ? Q.lam[r] * isf : 0.0;
  if (A.lam_g_out && NE > 0) {           // (scaled rows sg d(x): the multiplier of the user's row is sg times the scaled problem's)
    T.sync();
    for (int g = T.tid; g < nSl; g += T.nt) {
      const int e = g / NE1, i = g % NE1;
      if (!sh_cnt(A, mk_e(A, e))) continue;
      const int row = A.edge_row0[e] + NW + NX + i;
      A.lam_g_out[(int64_t)b * A.n_g + row] = Q.lam[row] * Q.sgn[e * NE1 + i] * isf;
    }
  }
  if (A.g_out) {
    // g in the reference's convention: equality rows = residual (+rhs 0), nl rows = d(x)
    for (int r = T.tid; r < A.n_g; r += T.nt) A.g_out[(int64_t)b * A.n_g + r] = sh_cnt(A, mk_g(A, r)) ? Q.c[r] : 0.0;
    T.sync();
    for (int g = T.tid; g < nSl; g += T.nt) {
      const int e = g / NE1, i = g % NE1;
      if (!sh_cnt(A, mk_e(A, e))) continue;
      const int row = A.edge_row0[e] + NW + NX + i;
      A.g_out[(int64_t)b * A.n_g + row] = (Q.c[row] + Q.s[e * NE1 + i]) / Q.sgn[e * NE1 + i];
    }
  }
  if (T.tid == 0) {
    if (A.f_out) A.f_out[b] = E.obj * isf;
    if (A.stats) {
      dompc_stats& S = A.stats[b];
      S.success = (status == 0 || status == 1) ? 1 : 0;
      S.status = status; S.iter_count = it; S.n_reg = n_reg; S.n_ls_fail = n_ls_fail; S.n_sweeps = n_sweeps; S.n_trials = n_trials; S.n_soc = n_soc; S.n_watchdog = n_watchdog; S.reserved0 = 0;
      S.mu = mu; S.obj = E.obj * isf; S.inf_pr = E.e_p; S.inf_du = E.e_d; S.inf_compl = comp_err(E.C, 0.0);
      S.obj_scaling = Q.sf; S.t_wall_total = 0.0;
    }
  }
  T.sync();
}

// ------------------------------------------------------------------------------------------------
// mode 1: one Newton direction at a given primal-dual point (parity tests against the oracle's
// sparse KKT solve).  Slacks: s = d(x) pushed inside, z_s = 1.
// (b: parameter vector / output row of a batched call - same point x, lam, z for every b; slot: workspace of the workgroup)
DOMPC_DEV inline void debug_newton(const Thr& T, const KArgs& A, int b = 0, int slot = 0) {
  const dompc_options& O = A.opt;
  Prob Q = make_prob(A, slot, A.p + (int64_t)b * A.n_opt_p);
  const int nX = A.n_opt_x;
  for (int g = T.tid; g < nX; g += T.nt) {
    Q.x[g] = A.x0[g]; Q.lb[g] = A.lbx[g]; Q.ub[g] = A.ubx[g];
    Q.zl[g] = A.dbg_zl[g]; Q.zu[g] = A.dbg_zu[g];
  }
  for (int r = T.tid; r < A.n_g; r += T.nt) Q.lam[r] = A.dbg_lam[r];
  T.sync();
  if (NE > 0) {
    for (int e = T.tid; e < A.n_edges; e += T.nt) {
      for (int i = 0; i < NE; ++i) { Q.s[e * NE1 + i] = 0.0; Q.sgn[e * NE1 + i] = 1.0; }
      if (DENSE_EDGE) dae_edge_f(Q, e, Q.x, Q.s, Q.ct); else eval_edge_f(Q, e, Q.x, Q.s, Q.ct);
      for (int i = 0; i < NE; ++i) {
        const int row = A.edge_row0[e] + NW + NX + i, si = e * NE1 + i;
        const double l = A.lbg[row], u = A.ubg[row];
        const bool hl = l > -INFINITY, hu = u < INFINITY;
        double pl = hl ? O.bound_push * fmax(1.0, fabs(l)) : 0.0;
        double pu = hu ? O.bound_push * fmax(1.0, fabs(u)) : 0.0;
        if (hl && hu) { pl = fmin(pl, O.bound_frac * (u - l)); pu = fmin(pu, O.bound_frac * (u - l)); }
        double sv = Q.ct[row];
        if (A.dbg_at_solution) {
          // a converged point: the row residual d(x) - s vanishes, the slack is strictly inside the bounds the solver
          // relaxed (bound_relax_factor), complementarity holds at the given barrier parameter
          const double lr = hl ? l - fmin(O.constr_viol_tol, O.bound_relax_factor * fmax(1.0, fabs(l))) : l;
          const double ur = hu ? u + fmin(O.constr_viol_tol, O.bound_relax_factor * fmax(1.0, fabs(u))) : u;
          const double tiny = 1e-12 * fmax(1.0, fabs(sv));
          if (hl) sv = fmax(sv, lr + tiny);
          if (hu) sv = fmin(sv, ur - tiny);
          Q.s[si] = sv; Q.sl[si] = lr; Q.su[si] = ur;
          Q.zsl[si] = hl ? A.dbg_mu / (sv - lr) : 0.0; Q.zsu[si] = hu ? A.dbg_mu / (ur - sv) : 0.0;
          continue;
        }
        if (hl) sv = fmax(sv, l + pl);
        if (hu) sv = fmin(sv, u - pu);
        Q.s[si] = sv; Q.sl[si] = l; Q.su[si] = u;
        Q.zsl[si] = hl ? 1.0 : 0.0; Q.zsu[si] = hu ? 1.0 : 0.0;
      }
    }
    T.sync();
  }
  Q.sf = 1.0;
  // (b, slot: the outlined phases rebuild their view of the problem from exactly these two - ADVICE r3: with the literal
  //  (0, 0) every workgroup of a batched call swept and factorised slot 0 with parameter row 0)
  run_sweep(T, Q, b, slot, A.dbg_mu, 0, A.dbg_delta);
  const int fail = run_backward(T, Q, b, slot, A.dbg_mu, A.dbg_delta);
  run_forward(T, Q, b, slot, A.dbg_mu, A.dbg_delta);
  for (int g = T.tid; g < nX; g += T.nt) {
    A.dbg_dx[(int64_t)b * nX + g] = fail ? NAN : Q.dx[g];
    A.dbg_rd[(int64_t)b * nX + g] = Q.rd[g];
  }
  for (int r = T.tid; r < A.n_g; r += T.nt) {
    A.dbg_dlam[(int64_t)b * A.n_g + r] = Q.dlam[r];
    A.dbg_c[(int64_t)b * A.n_g + r] = Q.c[r];
  }
  T.sync();
}

// number of doubles written per edge by the sweep kernel: [A|B] (NX*NA), c (NX), Qt (NA*NA), qv (NA)
constexpr int SWEEP_BLOCK = NX * NA + NX + NA * NA + NA;

// mode 2: model-evaluation sweep for a batch of iterates (one workgroup per iterate slot)
DOMPC_DEV inline void sweep_problem(const Thr& T, const KArgs& A, int b, int slot) {
  Prob Q = make_prob(A, slot, A.p + (int64_t)b * A.n_opt_p);
  const int nX = A.n_opt_x;
  const double* xin = A.sw_x + (int64_t)b * nX;
  const double* lin = A.sw_lam + (int64_t)b * A.n_g;
  for (int g = T.tid; g < nX; g += T.nt) {
    Q.x[g] = xin[g]; Q.lb[g] = -INFINITY; Q.ub[g] = INFINITY; Q.zl[g] = 0.0; Q.zu[g] = 0.0;
  }
  for (int r = T.tid; r < A.n_g; r += T.nt) Q.lam[r] = lin[r];
  for (int g = T.tid; g < A.n_edges * NE; g += T.nt) {
    const int si = (g / NE1) * NE1 + g % NE1;
    Q.s[si] = 0.0; Q.sl[si] = -INFINITY; Q.su[si] = INFINITY; Q.zsl[si] = 0.0; Q.zsu[si] = 0.0; Q.sgn[si] = 1.0;
  }
  T.sync();
  Q.sf = 1.0;
  run_sweep(T, Q, b, slot, 0.0);
  double* gout = A.sw_g + (int64_t)b * A.n_g;
  for (int r = T.tid; r < A.n_g; r += T.nt) gout[r] = Q.c[r];
  double* bl = A.sw_blocks + (int64_t)b * A.n_edges * SWEEP_BLOCK;
  for (int it = T.tid; A.sw_blocks && it < A.n_edges * SWEEP_BLOCK; it += T.nt) {
    const int e = it / SWEEP_BLOCK, i = it % SWEEP_BLOCK;
    const double* S_ = Q.ES(e);
    double v;
    if (i < NX * NA) v = S_[ES_AB + i];
    else if (i < NX * NA + NX) v = S_[ES_CV + i - NX * NA];
    else if (i < NX * NA + NX + NA * NA) v = S_[ES_QT + symi((i - NX * NA - NX) / NA, (i - NX * NA - NX) % NA, NA)];
    else v = S_[ES_QV + i - NX * NA - NX - NA * NA];
    bl[it] = v;
  }
  T.sync();
}

