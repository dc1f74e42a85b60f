// dompc_phases.h - structured interior-point solver, part of dompc_kernel.h (included there, inside namespace dompc, in this order:
// dompc_edge.h, dompc_factor.h, dompc_node.h, dompc_riccati.h, dompc_forward.h, dompc_sweep.h, dompc_phases.h, dompc_driver.h).
// Contents: thread-parallel vector passes (error measures, step rules, line search, accept) and the outlined phases of the device build.
// Sizes, record layouts, the thread context `Thr`, reductions and the small dense products are in dompc_kernel.h.

// ================================================================================================
// Outlined phases.  Inlined into one kernel, the phases share one register allocation: values that live across the
// whole IPM loop get spilled around the register-hungry phases and are reloaded from scratch at every use inside
// the hot loops of the others (measured: adding the matrix-core Riccati pass made the SWEEP 45 % slower).  As
// separate functions each phase has the whole register file; its context is rebuilt inside from uniform sources
// (kernel arguments from the kernarg segment, block / thread indices, v_readfirstlane of the few scalar arguments), so
// nothing is passed through memory.  Host emulation: plain calls.
// function-only evaluation of the trial point (line search): this thread's share of the objective; the constraint
// values of its edges go to Q.ct.  Straight-line model code with its own register allocation (inlined into the
// driver it was the main source of the driver's scratch traffic).
template <bool FINE>
DOMPC_DEV inline double trial_edges(const Thr& T, const Prob& Q) {
  const KArgs& A = *Q.A;
  double f = 0.0;
  if (FINE && !DENSE_EDGE && M > 0) {
    // one thread per piece of an edge; piece 0 of edge e (its objective share) on thread e like in the loop below: same partial sums
    constexpr int NPC = NI * DEG + 1;
    for (int it = T.tid; it < A.n_edges * NPC; it += T.nt) {
      const int e = it % A.n_edges, q = it / A.n_edges;
      const double fe = eval_edge_f_t<true>(Q, e, Q.xt, Q.st, Q.ct, q == 0 ? NI * DEG : q - 1);
      f += fe;
    }
  } else
  for (int e = T.tid; e < A.n_edges; e += T.nt) {
    const int m = mk_e(A, e);
    if (!m) continue;
    const double fe = DENSE_EDGE ? dae_edge_f(Q, e, Q.xt, Q.st, Q.ct) : eval_edge_f(Q, e, Q.xt, Q.st, Q.ct);
    if (sh_cnt(A, m)) f += fe;
  }
  for (int n = T.tid; n < A.n_nodes; n += T.nt)
    if (sh_cnt(A, mk_n(A, n))) f += node_rterm_f(Q, n, Q.xt);
  return f;
}
// ---- the thread-parallel passes of the line search (outlined on the device like the phases above: inlined into the
//      driver, their register arrays and the second call sites of the second-order correction cost the hot loops of
//      the driver 3 % in spills)
// largest ratios (-dx)/(x - l), dx/(u - x) and (-dz)/z over the bounded variables: the fraction-to-the-boundary steps are
// tau / ratio (one division at the end instead of one per bound), and the directional derivative of the barrier function
DOMPC_DEV inline void step_rules_pass(const Thr& T, const Prob& Q, double mu, double (&r5)[5]) {   // ratio_x, ratio_z, dphi of Q.dx / Q.ds
  const KArgs& A = *Q.A;
  const int nX = A.n_opt_x, nSl = A.n_edges * NE;
  for (int i = 0; i < 5; ++i) r5[i] = 0.0;
  {
    double x_[DOMPC_FW], l_[DOMPC_FW], u2_[DOMPC_FW], d_[DOMPC_FW], gf_[DOMPC_FW], zl_[DOMPC_FW], zu_[DOMPC_FW];
#define L_(u, g) x_[u] = Q.x[g]; l_[u] = Q.lb[g]; u2_[u] = Q.ub[g]; d_[u] = Q.dx[g]; gf_[u] = Q.gf[g]; zl_[u] = Q.zl[g]; zu_[u] = Q.zu[g];
#define B_(u, g)                                                                               \
    if (sh_cnt(A, mk_x(A, g))) {                                                           \
      const double xv = x_[u], l = l_[u], ub_ = u2_[u], d = d_[u];                         \
      double gphi = gf_[u];                                                                \
      if (l > -INFINITY) {                                                                 \
        const double r = fast_rcp(xv - l);                                                 \
        r5[0] = fmax(r5[0], -d * r);                            /* step to the bound */    \
        r5[1] = fmax(r5[1], 1.0 + r * d - mu * r * fast_rcp(zl_[u]));    /* -dz / z */     \
        gphi -= mu * r;                                                                    \
      }                                                                                    \
      if (ub_ < INFINITY) {                                                                \
        const double r = fast_rcp(ub_ - xv);                                               \
        r5[0] = fmax(r5[0], d * r);                                                        \
        r5[1] = fmax(r5[1], 1.0 - r * d - mu * r * fast_rcp(zu_[u]));                      \
        gphi += mu * r;                                                                    \
      }                                                                                    \
      if (KAPPA_D != 0.0) gphi += KAPPA_D * mu * one_sided(l, ub_);                        \
      r5[2] += gphi * d;                                                                   \
    }
    DOMPC_FOR4(nX, L_, B_)
#undef L_
#undef B_
  }
  for (int g = T.tid; g < nSl; g += T.nt) {
    if (!sh_cnt(A, mk_e(A, g / NE1))) continue;
    const int si = (g / NE1) * NE1 + g % NE1;
    const double sv = Q.s[si], l = Q.sl[si], u = Q.su[si], d = Q.ds[si];
    double gphi = 0.0;
    if (l > -INFINITY) {
      const double r = fast_rcp(sv - l);
      r5[0] = fmax(r5[0], -d * r);
      r5[1] = fmax(r5[1], 1.0 + r * d - mu * r * fast_rcp(Q.zsl[si]));
      gphi -= mu * r;
    }
    if (u < INFINITY) {
      const double r = fast_rcp(u - sv);
      r5[0] = fmax(r5[0], d * r);
      r5[1] = fmax(r5[1], 1.0 - r * d - mu * r * fast_rcp(Q.zsu[si]));
      gphi += mu * r;
    }
    if (KAPPA_D != 0.0) gphi += KAPPA_D * mu * one_sided(l, u);
    r5[2] += gphi * d;
  }
  const int ops[5] = {R_MAX, R_MAX, R_SUM, R_SUM, R_SUM};
  wg_reduce(T, r5, ops);
}
// objective, constraint violation and barrier sum of the trial point x + al * dx (left in Q.xt / Q.st, constraint values in Q.ct)
template <bool FINE>
DOMPC_DEV inline void eval_trial_pass(const Thr& T, const Prob& Q, double al, double& obj_o, double& th_o, double& bar_o) {
  const KArgs& A = *Q.A;
  const int nX = A.n_opt_x, nSl = A.n_edges * NE;
  double r3[3] = {0.0, 0.0, 0.0};    // obj, theta, barrier
  LogAcc La{1.0, 0, 0};
  double lin = 0.0;                  // distances to the single bound of the one-sided variables (damping term, KAPPA_D)
  {                                  // trial point and its barrier terms in one pass
    double x_[DOMPC_FW3], d_[DOMPC_FW3], l_[DOMPC_FW3], u2_[DOMPC_FW3];
#define L_(u, g) x_[u] = Q.x[g]; d_[u] = Q.dx[g]; l_[u] = Q.lb[g]; u2_[u] = Q.ub[g];
#define B_(u, g)                                                                               \
    if (mk_x(A, g)) {                                                                      \
      const double xt_ = x_[u] + al * d_[u];                                               \
      Q.xt[g] = xt_;                                                                       \
      if (sh_cnt(A, mk_x(A, g))) {                                                         \
        if (l_[u] > -INFINITY) logacc_add(La, xt_ - l_[u]);                                \
        if (u2_[u] < INFINITY) logacc_add(La, u2_[u] - xt_);                               \
        if (KAPPA_D != 0.0) { const double os_ = one_sided(l_[u], u2_[u]); lin += os_ > 0.0 ? xt_ - l_[u] : (os_ < 0.0 ? u2_[u] - xt_ : 0.0); } \
      }                                                                                    \
    }
    DOMPC_FORN(DOMPC_FW3, nX, L_, B_)
#undef L_
#undef B_
  }
  for (int g = T.tid; g < nSl; g += T.nt) {
    if (!mk_e(A, g / NE1)) continue;
    const int si = (g / NE1) * NE1 + g % NE1;
    Q.st[si] = Q.s[si] + al * Q.ds[si];
  }
  T.sync();
  for (int g = T.tid; g < NX; g += T.nt) Q.ct[g] = FREE_ROOT ? 0.0 : Q.xt[A.node_x_off[0] + g] - Q.P[g] / DOMPC_SX[g];
  if (FREE_ROOT && T.tid == 0) r3[0] += Q.sf * dompc_aterm_f(Q.xt + A.node_x_off[0], Q.P, Q.P + A.p_off_tvp, Q.P + A.p_off_p);
  r3[0] += trial_edges<FINE>(T, Q);
  T.sync();
  {
    double c_[DOMPC_FW1];
#define L_(u, g) c_[u] = Q.ct[g];
#define B_(u, g) if (sh_cnt(A, mk_g(A, g))) r3[1] += fabs(c_[u]);
    DOMPC_FORN(DOMPC_FW1, A.n_g, L_, B_)
#undef L_
#undef B_
  }
  for (int g = T.tid; g < nSl; g += T.nt) {
    if (!sh_cnt(A, mk_e(A, g / NE1))) continue;
    const int si = (g / NE1) * NE1 + g % NE1;
    if (Q.sl[si] > -INFINITY) logacc_add(La, Q.st[si] - Q.sl[si]);
    if (Q.su[si] < INFINITY) logacc_add(La, Q.su[si] - Q.st[si]);
    if (KAPPA_D != 0.0) { const double os_ = one_sided(Q.sl[si], Q.su[si]); lin += os_ > 0.0 ? Q.st[si] - Q.sl[si] : (os_ < 0.0 ? Q.su[si] - Q.st[si] : 0.0); }
  }
  r3[2] = -logacc_value(La);
  if (KAPPA_D != 0.0) r3[2] += KAPPA_D * lin;
  const int ops[3] = {R_SUM, R_SUM, R_SUM};
  wg_reduce(T, r3, ops);
  obj_o = r3[0]; th_o = r3[1]; bar_o = r3[2];
}
// the trial point becomes the iterate: x, s, bound multipliers (step a_z, safeguarded) and constraint multipliers (step alpha);
// returns this thread's complementarity statistics of the new iterate (consumed by measure() after the sweep)
DOMPC_DEV inline Comp accept_pass(const Thr& T, const Prob& Q, double alpha, double a_z, double mu) {
  const KArgs& A = *Q.A;
  const int nX = A.n_opt_x, nSl = A.n_edges * NE;
  const double ks = 1e10;
  Comp Cp{-INFINITY, INFINITY, 0.0};       // complementarity statistics of the new iterate (consumed by measure() after the sweep)
  {
    double xt_[DOMPC_FW], x_[DOMPC_FW], d_[DOMPC_FW], l_[DOMPC_FW], u2_[DOMPC_FW], zl_[DOMPC_FW], zu_[DOMPC_FW];
#define L_(u, g) xt_[u] = Q.xt[g]; x_[u] = Q.x[g]; d_[u] = Q.dx[g]; l_[u] = Q.lb[g]; u2_[u] = Q.ub[g]; zl_[u] = Q.zl[g]; zu_[u] = Q.zu[g];
#define B_(u, g)                                                                               \
    if (mk_x(A, g)) {                                                                        \
      const double xv = xt_[u], l = l_[u], ub_ = u2_[u];                                     \
      const bool cnt_ = sh_cnt(A, mk_x(A, g));                                               \
      Q.x[g] = xv;                                                                           \
      if (l > -INFINITY) {                                                                   \
        const double ro = fast_rcp(x_[u] - l);                     /* dz_lo with 1/(x - l) */  \
        const double z = zl_[u] + a_z * (mu * ro - zl_[u] - zl_[u] * ro * d_[u]);            \
        const double dd = xv - l, mr = mu * fast_rcp(dd);                                    \
        const double zn = fmax(fmin(z, ks * mr), mr * (1.0 / ks));                           \
        Q.zl[g] = zn;                                                                        \
        if (cnt_) comp_add(Cp, dd * zn, zn);                                                 \
      }                                                                                      \
      if (ub_ < INFINITY) {                                                                  \
        const double ro = fast_rcp(ub_ - x_[u]);                                             \
        const double z = zu_[u] + a_z * (mu * ro - zu_[u] + zu_[u] * ro * d_[u]);            \
        const double dd = ub_ - xv, mr = mu * fast_rcp(dd);                                  \
        const double zn = fmax(fmin(z, ks * mr), mr * (1.0 / ks));                           \
        Q.zu[g] = zn;                                                                        \
        if (cnt_) comp_add(Cp, dd * zn, zn);                                                 \
      }                                                                                      \
    }
    DOMPC_FOR4(nX, L_, B_)
#undef L_
#undef B_
  }
  for (int g = T.tid; g < nSl; g += T.nt) {
    if (!mk_e(A, g / NE1)) continue;
    const int si = (g / NE1) * NE1 + g % NE1;
    const double sv = Q.st[si], so = Q.s[si], dsv = Q.ds[si];
    const bool cnt_ = sh_cnt(A, mk_e(A, g / NE1));
    Q.s[si] = sv;
    const double l = Q.sl[si], u = Q.su[si];
    if (l > -INFINITY) {
      const double z = Q.zsl[si] + a_z * dz_lo(so, l, Q.zsl[si], dsv, mu);
      const double zn = fmax(fmin(z, ks * mu / (sv - l)), mu / (ks * (sv - l)));
      Q.zsl[si] = zn;
      if (cnt_) comp_add(Cp, (sv - l) * zn, zn);
    }
    if (u < INFINITY) {
      const double z = Q.zsu[si] + a_z * dz_up(so, u, Q.zsu[si], dsv, mu);
      const double zn = fmax(fmin(z, ks * mu / (u - sv)), mu / (ks * (u - sv)));
      Q.zsu[si] = zn;
      if (cnt_) comp_add(Cp, (u - sv) * zn, zn);
    }
  }
  {
    double y_[DOMPC_FW1], dy_[DOMPC_FW1];
#define L_(u, g) y_[u] = Q.lam[g]; dy_[u] = Q.dlam[g];
#define B_(u, g) if (mk_g(A, g)) Q.lam[g] = y_[u] + alpha * dy_[u];
    DOMPC_FORN(DOMPC_FW1, A.n_g, L_, B_)
#undef L_
#undef B_
  }
  return Cp;
}
struct PhaseRet { unsigned gen, nred, xseq; int rc; };
struct PhaseRet3 { unsigned gen, nred, xseq; double v0, v1, v2; };
#ifndef DOMPC_HOST_EMU
#define DOMPC_PHASE_PROLOGUE                                                        \
  const KArgs A = kernel_args(kp);                                                  \
  Thr T = make_thr(A);                                                              \
  hier_setup(T);                                                                    \
  T.kp = kp;                                                                        \
  T.gen = ufl(gen); T.nred = ufl(nred); T.xseq = ufl(xseq);                         \
  Prob Q = make_prob(A, ufl(slot), A.p + (int64_t)ufl(b) * A.n_opt_p);              \
  Q.sf = ufl(sf);
__device__ __attribute__((noinline)) PhaseRet phase_sweep(const void* kp, int b, int slot, double sf, double mu, double dsw, int soc, unsigned gen, unsigned nred, unsigned xseq) {
  DOMPC_PHASE_PROLOGUE
  Q.soc = ufl(soc);
  prob_bounds(Q);
  Q.dsw = ufl(dsw);
  const int rc = sweep<false>(T, Q, ufl(mu));
  return PhaseRet{T.gen, T.nred, T.xseq, rc};
}
__device__ __attribute__((noinline)) PhaseRet phase_sweep_fine(const void* kp, int b, int slot, double sf, double mu, double dsw, int soc, unsigned gen, unsigned nred, unsigned xseq) {
  DOMPC_PHASE_PROLOGUE
  Q.soc = ufl(soc);
  prob_bounds(Q);
  Q.dsw = ufl(dsw);
  const int rc = sweep<true>(T, Q, ufl(mu));
  return PhaseRet{T.gen, T.nred, T.xseq, rc};
}
__device__ __attribute__((noinline)) PhaseRet phase_backward(const void* kp, int b, int slot, double sf, double mu, double delta, double dsw, int mode, unsigned gen, unsigned nred, unsigned xseq) {
  DOMPC_PHASE_PROLOGUE
  Q.dsw = ufl(dsw);
  Q.soc = ufl(mode);
  prob_bounds(Q);
  const int rc = riccati_backward(T, Q, ufl(mu), ufl(delta));
  return PhaseRet{T.gen, T.nred, T.xseq, rc};
}
__device__ __attribute__((noinline)) PhaseRet phase_forward(const void* kp, int b, int slot, double sf, double mu, double delta, double dsw, unsigned gen, unsigned nred, unsigned xseq) {
  DOMPC_PHASE_PROLOGUE
  Q.dsw = ufl(dsw);
  riccati_forward_t<false>(T, Q, ufl(mu), ufl(delta));
  return PhaseRet{T.gen, T.nred, T.xseq, 0};
}
__device__ __attribute__((noinline)) PhaseRet phase_forward_adj(const void* kp, int b, int slot, double sf, double mu, double delta, double dsw, unsigned gen, unsigned nred, unsigned xseq) {
  DOMPC_PHASE_PROLOGUE
  Q.dsw = ufl(dsw);
  riccati_forward_t<true>(T, Q, ufl(mu), ufl(delta));
  return PhaseRet{T.gen, T.nred, T.xseq, 0};
}
__device__ __attribute__((noinline)) PhaseRet3 phase_step_rules(const void* kp, int b, int slot, double sf, double mu, unsigned gen, unsigned nred, unsigned xseq) {
  DOMPC_PHASE_PROLOGUE
  double r5[5];
  step_rules_pass(T, Q, ufl(mu), r5);
  return PhaseRet3{T.gen, T.nred, T.xseq, r5[0], r5[1], r5[2]};
}
__device__ __attribute__((noinline)) PhaseRet3 phase_eval_trial(const void* kp, int b, int slot, double sf, double al, unsigned gen, unsigned nred, unsigned xseq) {
  DOMPC_PHASE_PROLOGUE
  double o = 0.0, th = 0.0, br = 0.0;
  eval_trial_pass<false>(T, Q, ufl(al), o, th, br);
  return PhaseRet3{T.gen, T.nred, T.xseq, o, th, br};
}
__device__ __attribute__((noinline)) PhaseRet3 phase_eval_trial_fine(const void* kp, int b, int slot, double sf, double al, unsigned gen, unsigned nred, unsigned xseq) {
  DOMPC_PHASE_PROLOGUE
  double o = 0.0, th = 0.0, br = 0.0;
  eval_trial_pass<true>(T, Q, ufl(al), o, th, br);
  return PhaseRet3{T.gen, T.nred, T.xseq, o, th, br};
}
__device__ __attribute__((noinline)) PhaseRet3 phase_accept(const void* kp, int b, int slot, double sf, double alpha, double a_z, double mu, unsigned gen, unsigned nred, unsigned xseq) {
  DOMPC_PHASE_PROLOGUE
  const Comp C = accept_pass(T, Q, ufl(alpha), ufl(a_z), ufl(mu));
  return PhaseRet3{T.gen, T.nred, T.xseq, C.smax, C.smin, C.sum_z};
}
#undef DOMPC_PHASE_PROLOGUE
#define DOMPC_PHASE_CALL(fn, ...)                                                   \
  const auto r_ = fn(T.kp, b, slot, Q.sf, __VA_ARGS__, T.gen, T.nred, T.xseq);            \
  T.gen = ufl(r_.gen); T.nred = ufl(r_.nred); T.xseq = ufl(r_.xseq);
#endif
// measurement aid (tools/gpu_phase_traffic.sh): -DDOMPC_REPEAT_PHASE=<mask> runs the named phases TWICE per call.  They are idempotent
// (records and reductions of the same inputs), so iterates and iteration counts stay the same bits and the difference of two builds in
// time and in HBM traffic is the phase's own.  1 sweep, 2 backward + sweep (the backward pass works in place on what the sweep assembled), 4 forward (not the adjoint variant of the last levels: it reuses a slot of the node records), 8 trial evaluation, 16 step rules
#ifndef DOMPC_REPEAT_PHASE
#define DOMPC_REPEAT_PHASE 0
#endif
// dsw: the inertia correction this sweep folds into the condensed blocks; remembered in Q for the Riccati passes
DOMPC_DEV inline int run_sweep(const Thr& T, Prob& Q, int b, int slot, double mu, int soc = 0, double dsw = 0.0) {
  Q.dsw = dsw;
  Q.lu_ok = lu_store_rule(*Q.A, mu, soc) ? 1 : 0;
  Q.rp_soc = soc; Q.rp_mu = mu;
#ifndef DOMPC_HOST_EMU
  if (fine_items(T, *Q.A)) { DOMPC_PHASE_CALL(phase_sweep_fine, mu, dsw, soc) return ufl(r_.rc); }
  if (DOMPC_REPEAT_PHASE & 1) { DOMPC_PHASE_CALL(phase_sweep, mu, dsw, soc) }
  DOMPC_PHASE_CALL(phase_sweep, mu, dsw, soc)
  return ufl(r_.rc);
#else
  (void)b; (void)slot;
  Q.soc = soc;
  prob_bounds(Q);
  const int rc = fine_items(T, *Q.A) ? sweep<true>(T, Q, mu) : sweep<false>(T, Q, mu);
  Q.soc = 0;
  prob_bounds(Q);
  return rc;
#endif
}
// mode: Prob::soc of the sweep whose records the pass works on (only bit 1 matters here: objective Hessians left out)
DOMPC_DEV inline int run_backward(const Thr& T, const Prob& Q, int b, int slot, double mu, double delta, int mode = 0) {
#ifndef DOMPC_HOST_EMU
  if (DOMPC_REPEAT_PHASE & 2) {            // the pass turns the assembled node matrices into value functions in place: sweep again in between
    { DOMPC_PHASE_CALL(phase_backward, mu, delta, Q.dsw, mode) }
    { DOMPC_PHASE_CALL(phase_sweep, Q.rp_mu, Q.dsw, Q.rp_soc) }
  }
  DOMPC_PHASE_CALL(phase_backward, mu, delta, Q.dsw, mode)
  return ufl(r_.rc);
#else
  (void)b; (void)slot;
  Prob Qm = Q;
  Qm.soc = mode;
  prob_bounds(Qm);
  return riccati_backward(T, Qm, mu, delta);
#endif
}
DOMPC_DEV inline void run_step_rules(const Thr& T, const Prob& Q, int b, int slot, double mu, double (&r5)[5]) {
#ifndef DOMPC_HOST_EMU
  if (DOMPC_REPEAT_PHASE & 16) { DOMPC_PHASE_CALL(phase_step_rules, mu) }
  DOMPC_PHASE_CALL(phase_step_rules, mu)
  r5[0] = ufl(r_.v0); r5[1] = ufl(r_.v1); r5[2] = ufl(r_.v2); r5[3] = 0.0; r5[4] = 0.0;
#else
  (void)b; (void)slot;
  step_rules_pass(T, Q, mu, r5);
#endif
}
DOMPC_DEV inline void run_eval_trial(const Thr& T, const Prob& Q, int b, int slot, double al, double& obj_o, double& th_o, double& bar_o) {
#ifndef DOMPC_HOST_EMU
  if (fine_items(T, *Q.A)) { DOMPC_PHASE_CALL(phase_eval_trial_fine, al) obj_o = ufl(r_.v0); th_o = ufl(r_.v1); bar_o = ufl(r_.v2); return; }
  if (DOMPC_REPEAT_PHASE & 8) { DOMPC_PHASE_CALL(phase_eval_trial, al) }
  DOMPC_PHASE_CALL(phase_eval_trial, al)
  obj_o = ufl(r_.v0); th_o = ufl(r_.v1); bar_o = ufl(r_.v2);
#else
  (void)b; (void)slot;
  if (fine_items(T, *Q.A)) eval_trial_pass<true>(T, Q, al, obj_o, th_o, bar_o); else eval_trial_pass<false>(T, Q, al, obj_o, th_o, bar_o);
#endif
}
DOMPC_DEV inline Comp run_accept(const Thr& T, const Prob& Q, int b, int slot, double alpha, double a_z, double mu) {
#ifndef DOMPC_HOST_EMU
  DOMPC_PHASE_CALL(phase_accept, alpha, a_z, mu)
  return Comp{r_.v0, r_.v1, r_.v2};
#else
  (void)b; (void)slot;
  return accept_pass(T, Q, alpha, a_z, mu);
#endif
}
DOMPC_DEV inline void run_forward(const Thr& T, const Prob& Q, int b, int slot, double mu, double delta) {
#ifndef DOMPC_HOST_EMU
  if (forward_adjoint(Q, mu)) {
    DOMPC_PHASE_CALL(phase_forward_adj, mu, delta, Q.dsw)
  } else {
    if (DOMPC_REPEAT_PHASE & 4) { DOMPC_PHASE_CALL(phase_forward, mu, delta, Q.dsw) }
    DOMPC_PHASE_CALL(phase_forward, mu, delta, Q.dsw)
  }
#else
  (void)b; (void)slot;
  if (forward_adjoint(Q, mu)) riccati_forward_t<true>(T, Q, mu, delta);
  else riccati_forward_t<false>(T, Q, mu, delta);
#endif
}


