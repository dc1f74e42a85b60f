// dompc_sweep.h - structured interior-point solver, part of dompc_kernel.h (included there, inside namespace dompc, in this order:
// dompc_edge.h, dompc_factor.h, dompc_node.h, dompc_riccati.h, dompc_forward.h, dompc_sweep.h, dompc_phases.h, dompc_driver.h).
// Contents: derivative sweep at the current iterate: per-edge evaluation / condensing, node assembly, dummies.
// Sizes, record layouts, the thread context `Thr`, reductions and the small dense products are in dompc_kernel.h.

// ================================================================================================

// derivative sweep at the current iterate: per-edge evaluation/condensing, node assembly, dummies
// FINE: the thread-per-entry node assembly (a problem spread over several workgroups, fine_items) - its own instantiation and its own outlined
// phase, so that the code of the batch path is the one it was (sharing one function cost the batch path 3.5 % in a same-box A/B)
template <bool FINE>
DOMPC_DEV inline int sweep(const Thr& T, Prob& Q, double mu) {
  const KArgs& A = *Q.A;
  const int FSET = T.flag_begin(1);         // (every thread has read the previous sweep's verdict, see riccati_backward)
  long long pc0 = prof_clock();
#if DOMPC_PROFILE
#define DOMPC_PS(i) if (T.prof && T.tid == 0) { const long long pc1 = prof_clock(); T.prof[i] += pc1 - pc0; pc0 = pc1; }
#else
#define DOMPC_PS(i)
#endif
  (void)pc0;
  if (!Q.soc)
    for (int g = T.tid; g < NX; g += T.nt) Q.c[g] = FREE_ROOT ? 0.0 : Q.x[A.node_x_off[0] + g] - Q.P[g] / DOMPC_SX[g];
  if (FREE_ROOT && T.tid == 0) {
    // arrival cost of the free initial state (value, gradient, Hessian) into the root's node record
    double* at = Q.ND(0) + ND_AT;
    double hp[NX_T > 0 ? NX_T : 1], gr[NX > 0 ? NX : 1], val = 0.0;
    dompc_aterm(Q.x + A.node_x_off[0], Q.P, Q.P + A.p_off_tvp, Q.P + A.p_off_p, &val, gr, hp);
    const double wh = (Q.soc & 2) ? 0.0 : Q.sf;
    at[0] = Q.sf * val;
    for (int a = 0; a < NX; ++a) at[1 + a] = Q.sf * gr[a];
    for (int a = 0; a < NX; ++a)
      for (int b = 0; b < NX; ++b) at[1 + NX + a * NX + b] = wh * hp[symi(a, b, NX)];
  }
  if (!(DOMPC_KO & 8)) eval_models(T, Q);
  T.sync();
  DOMPC_PS(21)
  for (int rep = 0; rep < A.trace_pad; ++rep) {      // measurement aid (DOMPC_EXTRA_TRAFFIC): extra read+write passes over the model-output records
    for (int i = T.tid; i < A.n_edges * MO_REC; i += T.nt) { volatile double* p_ = Q.mo + i; *p_ = *p_; }
    T.sync();
  }
#ifndef DOMPC_HOST_EMU
  if constexpr (QUAD_EDGE) {
    if (sweep_quads(T, Q, mu)) T.fset(1, FSET);
  } else
#endif
  {
    const int ng = T.nt / T.gs, gid = group_index(T.tid, T.gs), lane = T.tid % T.gs;
    ldsd* Ld = T.edge_lds + (int64_t)(T.ltid / T.gs) * EL_SIZE;
    const int rounds = (A.n_edges + ng - 1) / ng;
    int staged_e = -1;
    const MocMap mm = moc_map(lane, T.gs);
    if (MO_COMPACT) mo_image_init(Ld + EL_MOS, lane, T.gs);
    if (MFMA_GJ) gj_table_init(Ld, lane);            // (MFMA_GJ: one wavefront per edge group)
    for (int rd = 0; rd < rounds; ++rd) {
      const int e = rd * ng + gid;
      const int en = e + ng;
      const bool mine = e < A.n_edges && mk_e(A, e);
      if (sh_on(A) && !mine) continue;                  // sharded: another rank's edge (no workgroup barrier inside)
      if constexpr (DENSE_EDGE) {
        static_assert(!DENSE_EDGE || dae::DG_SIZE == DAE_NEED, "LDS working set of the dense DAE path");
        if (eval_edge_dae(T, Q, mine ? e : -1, mu, lane, T.gs, Ld)) T.fset(1, FSET);
        continue;
      }
      if (eval_edge_coop(T, Q, mine ? e : -1, (en < A.n_edges && mk_e(A, en)) ? en : -1, mu, lane, T.gs, Ld, staged_e, mm)) T.fset(1, FSET);
    }
  }
  T.sync();
  DOMPC_PS(22)
  if (FINE) {
    constexpr int NVN = NX + NU + NS;
    for (int it = T.tid; it < A.n_nodes * NVN; it += T.nt) assemble_entry(Q, it / NVN, it % NVN);
  } else
  for (int n = T.tid; n < A.n_nodes; n += T.nt) {
    if (!mk_n(A, n)) continue;
    const int ci = cut_of(A, n);
    if (ci >= 0) assemble_children(Q, n, true, A.xbuf + x_asm(A) + ci * ASM_N);   // completed after the exchange
    else assemble_node(Q, n);
  }
  for (int d = T.tid; d < A.n_dummy; d += T.nt) {
    const int g = A.dummy_idx[d];
    Q.gf[g] = 0.0;
    Q.rd[g] = -Q.zl[g] + Q.zu[g];
  }
  T.sync();
  DOMPC_PS(23)
  if (sh_on(A)) {
    // cut parents: sum the child-dependent parts over the ranks; the failure flag rides along
    double* fl = A.xbuf + x_asm(A) + A.n_cut * ASM_N;
    for (int w = T.tid; w < A.shard_world; w += T.nt) fl[w] = (w == A.shard_rank && (T.fget(1) == FSET)) ? 1.0 : 0.0;
    T.xchg(x_asm(A), A.n_cut * ASM_N + A.shard_world);
    const int n0 = A.level_node_start[A.cut_level - 1];
    for (int ci = T.tid; ci < A.n_cut; ci += T.nt) assemble_finish(Q, n0 + ci, A.xbuf + x_asm(A) + ci * ASM_N);
    int bad = 0;
    for (int w = 0; w < A.shard_world; ++w) bad |= (fl[w] != 0.0);
    T.sync();
    return bad;
  }
  return (T.fget(1) == FSET);
}

// Barrier-parameter change at an unchanged iterate: only the barrier gradients move, linearly in mu.
// Updates the mu-dependent pieces of the per-edge records (rw, the condensed gradient W'rw, the slack
// residual) instead of repeating the whole derivative sweep.
DOMPC_PHASE void refresh_mu(const Thr& T, const Prob& Q, double dmu) {
  const KArgs& A = *Q.A;
  const int GS = T.gs, ng = T.nt / GS, gid = group_index(T.tid, GS), lane = T.tid % GS;
  ldsd* Ld = T.edge_lds + (int64_t)(T.ltid / GS) * EL_SIZE;
  for (int e = gid; e < A.n_edges; e += ng) {
    if (!mk_e(A, e)) continue;
    if (NW > 0) {
      const int woff = A.edge_w_off[e], zoff = (NZ > 0) ? edge_zoff(A, e) : 0;
      for (int r = lane; r < NW; r += GS) {
        const int gi = wvar(woff, zoff, r);
        Q.EW(e, EW_RW + r) += dmu * bar_grad(Q.x[gi], Q.lb[gi], Q.ub[gi], 1.0);
      }
      double* S_ = Q.ES(e);
      for (int a = lane; a < NA; a += GS) S_[ES_QV + a] += dmu * S_[ES_QVB + a];      // (W'b was formed by the sweep)
    }
    if (NE > 0) {
      double* S_ = Q.ES(e);
      for (int i = lane; i < NE; i += GS) {
        const int si = e * NE1 + i;
        S_[ES_RSN + i] += dmu * bar_grad(Q.s[si], Q.sl[si], Q.su[si], 1.0);
      }
    }
  }
  T.sync();
}

// bound multiplier steps of the primal-dual system:  dz_L = mu/(x-l) - z_L - z_L/(x-l) dx ,  dz_U = mu/(u-x) - z_U + z_U/(u-x) dx
DOMPC_DEV inline double dz_lo(double x, double l, double z, double d, double mu) { return mu / (x - l) - z - z / (x - l) * d; }
DOMPC_DEV inline double dz_up(double x, double u, double z, double d, double mu) { return mu / (u - x) - z + z / (u - x) * d; }

// Sum of logarithms of many positive numbers with ONE log(): the mantissas are multiplied, the exponents added
// (frexp: two instructions on the device) - sum log a_i = log(prod frac_i) + (sum exp_i) ln 2.  A non-positive or NaN
// term makes the sum NaN, as log() would (a trial point outside its bounds must fail the line search).
// The barrier terms of the line search cost ~100 instructions per variable and bound with log().
struct LogAcc { double m; int e; int bad; };
DOMPC_DEV inline void logacc_add(LogAcc& L, double a) {
  if (!(a > 0.0) || !(a < INFINITY)) L.bad = 1;
  int ea = 0;
  const double fa = frexp(a, &ea);
  L.m *= fa;
  L.e += ea;
  if (L.m < 0x1p-500) { int em = 0; L.m = frexp(L.m, &em); L.e += em; }
}
DOMPC_DEV inline double logacc_value(const LogAcc& L) { return L.bad ? NAN : log(L.m) + (double)L.e * 0.6931471805599453; }

// Complementarity statistics of the bounded variables: extremes of the products s = (x-l) z_L, (u-x) z_U and the
// sum of the multipliers.  max_i |s_i - mu| = max(s_max - mu, mu - s_min) gives the complementarity error for ANY
// barrier parameter without another pass over the variables (the barrier-update test needs it at several mu).
struct Comp { double smax, smin, sum_z; };      // thread-local partials or reduced values
DOMPC_DEV inline void comp_add(Comp& C, double s, double z) { C.smax = fmax(C.smax, s); C.smin = fmin(C.smin, s); C.sum_z += z; }
DOMPC_DEV inline double comp_err(const Comp& C, double mu) { return C.smax >= C.smin ? fmax(C.smax - mu, mu - C.smin) : 0.0; }

// Strided loop over [0, n) by the threads of the problem, DOMPC_FW elements per thread and trip: LOAD(u, g) pulls the
// operands of element g into slot u (all loads of a trip are issued before anything is computed from them - a plain
// grid-stride loop keeps ONE dependent load -> compute -> store chain per thread in flight and spends its time
// waiting for HBM), BODY(u, g) consumes slot u.
#ifndef DOMPC_FW
#define DOMPC_FW 8                     // elements per thread and trip (measured on MI355X, industrial_poly B = 4096: 4 -> 8 -2 % total time, 16 another -1.5 %)
#endif
// Round 4 experiment: the width per LOOP (DOMPC_FORN, -DDOMPC_FW_TUNED=1).  A trip is one dependent memory round trip of the wavefront,
// and with one wavefront per problem a pass over an iterate-sized vector is 14 - 16 of them at 8 elements per thread; loops that read
// one or two arrays afford 32 elements per thread in the same registers (4 trips), four arrays 16 - 60 instead of 106 trips per
// iteration over the four vector phases.  Every thread still visits its elements (g = tid mod nt) in increasing order: results bit
// for bit the same.  Measured (same box, interleaved): 6 176 / 6 179 vs 6 205 / 6 173 steps/s at B = 4096, 6 592 vs 6 602 at 16 384 -
// nothing: these passes are not bound by their round trips but by the bytes (the memory system as a whole moves ~2.8 TB/s with this
// access mix), so only fewer bytes would shorten them.  Off by default.
#ifndef DOMPC_FW_TUNED
#define DOMPC_FW_TUNED 0
#endif
#if DOMPC_FW_TUNED && !defined(DOMPC_HOST_EMU)
#define DOMPC_FW1 32                   // loops over one or two arrays
#define DOMPC_FW3 16                   // three or four arrays
#else
#define DOMPC_FW1 DOMPC_FW
#define DOMPC_FW3 DOMPC_FW
#endif
#define DOMPC_FOR4(n, LOAD, BODY) DOMPC_FORN(DOMPC_FW, n, LOAD, BODY)
#define DOMPC_FORN(FW_, n, LOAD, BODY)                                         \
  for (int g0_ = T.tid; g0_ < (n); g0_ += (FW_) * T.nt) {                      \
    _Pragma("unroll") for (int u_ = 0; u_ < (FW_); ++u_) {                     \
      const int g_ = g0_ + u_ * T.nt;                                          \
      const int gc_ = g_ < (n) ? g_ : g0_;                                     \
      LOAD(u_, gc_)                                                            \
    }                                                                          \
    _Pragma("unroll") for (int u_ = 0; u_ < (FW_); ++u_) {                     \
      const int g_ = g0_ + u_ * T.nt;                                          \
      if (g_ < (n)) { BODY(u_, g_) }                                           \
    }                                                                          \
  }

// error measures (IPOPT eq. (5)/(6)) + objective + theta at the current iterate.  `pre`: thread-local complementarity
// partials already accumulated by the caller (the accept pass has the updated x, z in registers), or null.
struct Errs { double e_d, e_p, sum_y, obj, theta; Comp C; };
DOMPC_PHASE Errs measure(const Thr& T, const Prob& Q, const Comp* pre) {
  const KArgs& A = *Q.A;
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // e_d, e_p, sum|y|, obj, theta, smax, -smin, sum z
  Comp C = pre ? *pre : Comp{-INFINITY, INFINITY, 0.0};
  if (pre) {
    double rd_[DOMPC_FW3];
#define L_(u, g) rd_[u] = Q.rd[g]; if (KAPPA_D != 0.0) rd_[u] += KAPPA_D * Q.mu * one_sided(Q.lb[g], Q.ub[g]);
#define B_(u, g) if (sh_cnt(A, mk_x(A, g))) v[0] = fmax(v[0], fabs(rd_[u]));
    DOMPC_FORN(DOMPC_FW3, A.n_opt_x, L_, B_)
#undef L_
#undef B_
  } else {
    double rd_[DOMPC_FW], x_[DOMPC_FW], l_[DOMPC_FW], u2_[DOMPC_FW], zl_[DOMPC_FW], zu_[DOMPC_FW];
#define L_(u, g) rd_[u] = Q.rd[g]; x_[u] = Q.x[g]; l_[u] = Q.lb[g]; u2_[u] = Q.ub[g]; zl_[u] = Q.zl[g]; zu_[u] = Q.zu[g];
#define B_(u, g)                                                                   \
    if (sh_cnt(A, mk_x(A, g))) {                                                   \
      v[0] = fmax(v[0], fabs(rd_[u] + (KAPPA_D != 0.0 ? KAPPA_D * Q.mu * one_sided(l_[u], u2_[u]) : 0.0)));  \
      if (l_[u] > -INFINITY) comp_add(C, (x_[u] - l_[u]) * zl_[u], zl_[u]);        \
      if (u2_[u] < INFINITY) comp_add(C, (u2_[u] - x_[u]) * zu_[u], zu_[u]);       \
    }
    DOMPC_FOR4(A.n_opt_x, L_, B_)
#undef L_
#undef B_
  }
  for (int g = T.tid; g < A.n_edges * NE; g += T.nt) {
    const int e = g / NE1, i = g % NE1;
    if (!sh_cnt(A, mk_e(A, e))) continue;
    const int si = e * NE1 + i;
    const double yd = Q.lam[A.edge_row0[e] + NW + NX + i];
    v[0] = fmax(v[0], fabs(-yd - Q.zsl[si] + Q.zsu[si] + (KAPPA_D != 0.0 ? KAPPA_D * Q.mu * one_sided(Q.sl[si], Q.su[si]) : 0.0)));
    if (!pre) {
      const double l = Q.sl[si], u = Q.su[si];
      if (l > -INFINITY) comp_add(C, (Q.s[si] - l) * Q.zsl[si], Q.zsl[si]);
      if (u < INFINITY) comp_add(C, (u - Q.s[si]) * Q.zsu[si], Q.zsu[si]);
    }
  }
  {
    double c_[DOMPC_FW1], y_[DOMPC_FW1];
#define L_(u, g) c_[u] = Q.c[g]; y_[u] = Q.lam[g];
#define B_(u, g) if (sh_cnt(A, mk_g(A, g))) { v[1] = fmax(v[1], fabs(c_[u])); v[2] += fabs(y_[u]); v[4] += fabs(c_[u]); }
    DOMPC_FORN(DOMPC_FW1, A.n_g, L_, B_)
#undef L_
#undef B_
  }
  for (int e = T.tid; e < A.n_edges; e += T.nt)
    if (sh_cnt(A, mk_e(A, e))) v[3] += Q.ES(e)[ES_OBJ];
  for (int n = T.tid; n < A.n_nodes; n += T.nt)
    if (sh_cnt(A, mk_n(A, n))) v[3] += node_rterm_f(Q, n, Q.x);
  if (FREE_ROOT && T.tid == 0) v[3] += Q.ND(0)[ND_AT];
  v[5] = C.smax; v[6] = -C.smin; v[7] = C.sum_z;
  const int ops[8] = {R_MAX, R_MAX, R_SUM, R_SUM, R_SUM, R_MAX, R_MAX, R_SUM};
  wg_reduce(T, v, ops);
  Errs E;
  E.e_d = v[0]; E.e_p = v[1]; E.sum_y = v[2]; E.obj = v[3]; E.theta = v[4];
  E.C.smax = v[5]; E.C.smin = -v[6]; E.C.sum_z = v[7];
  return E;
}

