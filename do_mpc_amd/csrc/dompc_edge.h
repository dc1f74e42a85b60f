// dompc_edge.h - structured interior-point solver, part of dompc_kernel.h (included there, inside namespace dompc, in this order:
// dompc_edge.h, dompc_factor.h, dompc_node.h, dompc_riccati.h, dompc_forward.h, dompc_sweep.h, dompc_phases.h, dompc_driver.h).
// Contents: trial evaluation of an edge (no derivatives); derivative evaluation + condensing of an edge (generic path); dense image of the compact model-output record; lane broadcast helpers.
// Sizes, record layouts, the thread context `Thr`, reductions and the small dense products are in dompc_kernel.h.

// ================================================================================================
// Trial evaluation: constraint residuals + objective share of one edge at `xv` (no derivatives).
// nlp_g / nlp_f of the reference for the rows/terms owned by edge e.
// SPLIT: one piece of the edge per call - `part` in [0, NI * DEG): the rows of that collocation point; NI * DEG: everything else (element /
// node continuity rows, objective share, nl_cons rows).  A single problem spread over many workgroups (wide mode) has far more threads than
// edges: the trial evaluation of the line search then runs one thread per piece instead of one per edge (trial_edges).
template <bool SPLIT>
DOMPC_DEV inline double eval_edge_f_t(const Prob& Q, int e, const double* xv, const double* sv, double* cv, int part) {
  constexpr int REST = NI * DEG;
  const KArgs& A = *Q.A;
  const int n = A.edge_parent[e], cn = A.edge_child[e], k = A.edge_level[e];
  const double* xn = xv + A.node_x_off[n];
  const double* un = xv + A.node_u_off[n];
  const double* xc = xv + A.node_x_off[cn];
  const double* w = xv + A.edge_w_off[e];
  const double* pp = Q.P + A.p_off_p + A.edge_pidx[e] * NP;
  const double* tvp = Q.P + A.p_off_tvp + k * NTVP;
  const int row0 = A.edge_row0[e];
  const double om = A.edge_omega[e] * Q.sf;
  double f[NX];
  // Round 6: the residual rows of a collocation point (and the continuity / end-point rows of an element) are formed in registers and
  // stored together, behind the loads they need.  Stored one by one in the loop that forms them, every load of the next row waited for
  // the store in front of it to reach memory (the compiler cannot move a load over a store that may alias; s_waitcnt vmcnt is in order):
  // 30 write round trips per edge and trial point instead of 3, ~300 k of the ~700 k cycles of a trial evaluation.  Same arithmetic.
  if (M == 0) {
    dompc_dyn_f(xn, un, nullptr, tvp, pp, f);
    double rl[NX];
    for (int a = 0; a < NX; ++a) rl[a] = f[a] - xc[a];
    for (int a = 0; a < NX; ++a) cv[row0 + a] = rl[a];
  } else {
    (void)REST;
    for (int i = 0; i < NI; ++i) {
      const double* xi0 = (i == 0) ? xn : w + slot_of(i, 0) * NX;
      const int rb = row0 + i * (DEG + 1) * NX;
      for (int j = 1; j <= DEG; ++j) {
        if (SPLIT && part != i * DEG + (j - 1)) continue;
        const double* xij = w + slot_of(i, j) * NX;
        dompc_dyn_f(xij, un, nullptr, tvp, pp, f);
        double rl[NX];
#pragma unroll
        for (int a = 0; a < NX; ++a) {
          double xp = DOMPC_C[0 * (DEG + 1) + j] * xi0[a];
          for (int r = 1; r <= DEG; ++r) xp += DOMPC_C[r * (DEG + 1) + j] * w[slot_of(i, r) * NX + a];
          rl[a] = f[a] - xp;
        }
#pragma unroll
        for (int a = 0; a < NX; ++a) cv[rb + (j - 1) * NX + a] = rl[a];
      }
      if (SPLIT && part != REST) continue;
      const double* xnext = w + next_slot(i) * NX;
      double rl[NX];
#pragma unroll
      for (int a = 0; a < NX; ++a) {
        double xf = DOMPC_D[0] * xi0[a];
        for (int r = 1; r <= DEG; ++r) xf += DOMPC_D[r] * w[slot_of(i, r) * NX + a];
        rl[a] = xnext[a] - xf;
      }
#pragma unroll
      for (int a = 0; a < NX; ++a) cv[rb + DEG * NX + a] = rl[a];
    }
    if (SPLIT && part != REST) return 0.0;
    double rl[NX];
#pragma unroll
    for (int a = 0; a < NX; ++a) rl[a] = w[(M - 1) * NX + a] - xc[a];
#pragma unroll
    for (int a = 0; a < NX; ++a) cv[row0 + NW + a] = rl[a];
  }
  double obj = om * lterm_f_e(Q, e, xn, un, nullptr, tvp, pp);
  if (k == A.N - 1) obj += om * mterm_f_e(Q, e, xc, Q.P + A.p_off_tvp + (k + 1) * NTVP, pp);
  if (RT_CUSTOM) obj += edge_rterm_f(Q, e, xv);
  if (NE > 0) {
    double d[NE1];
    nlcons_f_e(Q, e, xn, un, nullptr, tvp, pp, d);
    const double* eps = (NSE > 0) ? xv + A.node_eps_off[n] : nullptr;
    for (int i = 0; i < NE; ++i) {
      if (nl_slack(i) >= 0) d[i] -= eps[nl_slack(i)];
      d[i] *= Q.sgn[e * NE1 + i];                       // (constraint scaling of the row, solve_problem)
      cv[row0 + NW + NX + i] = d[i] - sv[e * NE1 + i];
    }
    for (int q = 0; q < NSE; ++q) obj += Q.sf * DOMPC_EPS_PEN[q] * eps[q];
  }
  return obj;
}
DOMPC_PHASE double eval_edge_f(const Prob& Q, int e, const double* xv, const double* sv, double* cv) { return eval_edge_f_t<false>(Q, e, xv, sv, cv, -1); }

// rterm share of node n (all outgoing edges): sum_b omega_k r'(u_n - u_prev)^2  (_mpc.py:1271-1275)
DOMPC_DEV inline const double* uprev_ptr(const Prob& Q, int n, const double* xv, double* tmp) {
  const KArgs& A = *Q.A;
  const int pn = A.node_parent[n];
  if (pn >= 0) return xv + A.node_u_off[pn];
  for (int i = 0; i < NU; ++i) tmp[i] = Q.P[A.p_off_uprev + i] / DOMPC_SU[i];
  return tmp;
}
DOMPC_DEV inline double node_rweight(const Prob& Q, int n) {
  const KArgs& A = *Q.A;
  const int cc = A.node_child_count[n];
  return cc > 0 ? cc * A.edge_omega[A.node_child_start[n]] * Q.sf : 0.0;
}
DOMPC_DEV inline double node_rterm_f(const Prob& Q, int n, const double* xv) {
  const KArgs& A = *Q.A;
  if (RT_CUSTOM) return 0.0;                       // (user-defined rterm: part of the edges' objective shares, edge_rterm_f)
  if (A.node_u_off[n] < 0) return 0.0;
  double tmp[NU];
  const double* up = uprev_ptr(Q, n, xv, tmp);
  const double* u = xv + A.node_u_off[n];
  const double rw = node_rweight(Q, n);
  double v = 0.0;
  for (int i = 0; i < NU; ++i) v += rw * DOMPC_RTERM[i] * (u[i] - up[i]) * (u[i] - up[i]);
  return v;
}

// user-defined rterm of edge e (parent node n): omega_k rterm(x_n, u_n, u_prev, tvp_k, p_e) with x, u unscaled inside the
// generated function and u_prev SCALED (_mpc.py:1263-1269)
DOMPC_DEV inline double edge_rterm_f(const Prob& Q, int e, const double* xv) {
  const KArgs& A = *Q.A;
  const int n = A.edge_parent[e];
  double tmp[NU > 0 ? NU : 1];
  const double* up = uprev_ptr(Q, n, xv, tmp);
  return A.edge_omega[e] * Q.sf * dompc_rterm_f(xv + A.node_x_off[n], xv + A.node_u_off[n], up, Q.P + A.p_off_tvp + A.edge_level[e] * NTVP,
                                                Q.P + A.p_off_p + A.edge_pidx[e] * NP);
}
// ... with derivatives, weighted (Hessian: zero in the least-squares multiplier solve, Prob::soc bit 1), by ONE lane into
// dst[0 .. RT_LEN): value, gradient over (x, u, u_prev), packed Hessian
DOMPC_DEV inline void edge_rterm_eval(const Prob& Q, int e, ldsd* dst) {
  const KArgs& A = *Q.A;
  const int n = A.edge_parent[e];
  const double om = A.edge_omega[e] * Q.sf, omh = (Q.soc & 2) ? 0.0 : om;
  double tmp[NU > 0 ? NU : 1], out[RT_LEN > 0 ? RT_LEN : 1];
  const double* up = uprev_ptr(Q, n, Q.x, tmp);
  dompc_rterm(Q.x + A.node_x_off[n], Q.x + A.node_u_off[n], up, Q.P + A.p_off_tvp + A.edge_level[e] * NTVP,
              Q.P + A.p_off_p + A.edge_pidx[e] * NP, out, out + 1, out + 1 + NR);
  for (int i = 0; i < 1 + NR; ++i) dst[i] = om * out[i];
  for (int i = 0; i < NR_T; ++i) dst[1 + NR + i] = omh * out[1 + NR + i];
}
// record part: d/d u_prev and the Hessian go to the node level (assembly, Riccati recursion)
DOMPC_DEV inline void edge_rterm_store(const ldsd* src, double* S_, int lane, int GS) {
  for (int i = lane; i < NU; i += GS) S_[ES_RTUP + i] = src[1 + NA + i];
  for (int i = lane; i < NR_T; i += GS) S_[ES_RTH + i] = src[1 + NR + i];
}

// ================================================================================================
// Derivative evaluation + condensing of one edge, cooperatively by a group of GS lanes (one wavefront
// on the device, one thread in the host emulation) with the edge's working set in LDS:
//   Mx = [G_w | G_y | r_g]  (NW x (NW+NA+1)) is built from the per-point model Jacobians, then inverted
//   in place by Gauss-Jordan elimination with partial pivoting (every elimination step updates all
//   NW x NC entries -> evenly spread over the lanes).  Afterwards the first NW columns hold G_w^-1
//   (kept for the multiplier recovery), the rest -W and -w0.
// All groups of the workgroup run this function in lock step (same trip counts), so the block-level
// barrier T.sync() is safe; groups with e < 0 only take part in the barriers.
constexpr int NC = NW + NA + 1;
static_assert(NW <= 128, "collocation block larger than 128 unknowns per edge is not supported (pivot key / used mask of the in-LDS elimination)");
static_assert(NI >= 2 || NW <= 64, "single finite element: at most 64 unknowns per edge (one extended column per lane of the register-resident elimination)");
static_assert(!DENSE_EDGE || NW <= 64, "dense edge path (algebraic states, rows at the collocation points, estimators): at most 64 unknowns per edge (one row per lane in its pivot search)");
// (single finite element: the matrix is assembled and eliminated in registers, LDS only holds W | w0 afterwards)
constexpr int MX_LD = (NI == 1) ? NA + 1 : NC;                         // leading dimension of the LDS matrix
constexpr int MX_W = (NI == 1) ? 0 : NW;                               // column offset of [W | w0] inside it
#ifndef DOMPC_HOST_EMU
constexpr bool TILE_CONDENSE = (NI == 1) && (DEG >= 1) && (NA <= 16) && !DENSE_EDGE;   // condensing on the matrix cores with register tiles (eval_edge_coop)
#else
constexpr bool TILE_CONDENSE = false;
#endif
// (the regions of the LDS-staged generic condensing - T1 beyond its first NW entries, U1, HUU, QT/HP - do not exist in the
//  matrix-core variant: 750 doubles per wavefront for industrial_poly)
constexpr int EL_MX = 0;
// (blocked elimination on the matrix cores, edge_factor_mfma: the W | w0 region doubles as its panel buffer - 4 columns of the
//  padded collocation block - plus one row of 64 dual-residual products)
constexpr int GJ_LDS = (NI == 1 && DEG >= 1) ? 4 * (((DEG * NX + 3) / 4) * 4) + 256 : 0;
constexpr int EL_T1 = EL_MX + (NW * MX_LD > GJ_LDS ? NW * MX_LD : GJ_LDS);   // Hww W  (NW x NA); first NW entries: the residual rows
constexpr int EL_T0 = EL_T1 + (TILE_CONDENSE ? NW : NW * NA);          // Hww w0 (NW)
constexpr int EL_RW = EL_T0 + NW;                                      // Newton-form gradient of w (NW)
constexpr int EL_SG = EL_RW + NW;                                      // Sigma_w (NW)
constexpr int EL_BB = EL_SG + NW;                                      // barrier gradient of w per unit mu (NW)
constexpr int EL_QV = EL_BB + NW;                                      // q~ (NA) and W'b (NA): stored by phase 7 (q~ together with r_y)
constexpr int EL_U1 = EL_QV + 2 * NA;                                  // Huw W (NU x NA), Huw w0 (NU)
constexpr int EL_HUU = EL_U1 + (TILE_CONDENSE ? 0 : NU * NA + NU);     // sum_p Huu_p (NU x NU)
constexpr int EL_QT = EL_HUU + (TILE_CONDENSE ? 0 : NU * NU);          // W'T1 (NA x NA), W'W (NA x NA)
constexpr int EL_HP = EL_QT;                                           // staged point Hessians H_p (NA x NA each): dead before QT is written
constexpr int EL_NHP = TILE_CONDENSE ? 0 : (NI * DEG > 2 ? NI * DEG : 2);
constexpr int EL_PV = EL_QT + EL_NHP * NA * NA;                        // pivot rows (NW)
constexpr int EL_RY = EL_PV + NW;                                      // G_y' lambda (NA), completed in phase 7
constexpr int EL_RT = EL_RY + NA;                                      // user-defined rterm of the edge: value, gradient, Hessian (RT_LEN)
#ifndef DOMPC_R16_NL
#define DOMPC_R16_NL 1                 // matrix-core Riccati pass also for models with nl_cons rows / slack variables
#endif
// (a user-defined rterm expression is not supported by tree sharding: the cut-parent update keeps the analytic form;
//  MPC.shard_tree refuses it)
constexpr bool R16_ENABLED = (NYT <= 16) && (NV <= 4) && (DOMPC_R16_NL ? (NE <= 4) : (NE == 0 && NS == 0)) && (DOMPC_SHARD == 0) && !RT_CUSTOM && !FREE_ROOT;   // dompc_riccati16.h (device)
#ifndef DOMPC_HOST_EMU
constexpr bool RB_IN_LDS = !R16_ENABLED;
#else
constexpr bool RB_IN_LDS = true;
#endif
constexpr int RB_NEED = RB_IN_LDS ? 2 * (NYT * NYT + NYT) + 5 * NA * NA + 6 * NA + NV * NA + NV + NE * (NA + 4) : 0;   // = rb::RB_SIZE (asserted there)
// forward pass: step vectors + staged operands of a chain-node step (riccati_forward); matrix-core Riccati: two staging buffers
constexpr int RF_NEED = 3 * NA + NV + NX + 3 * NW1 + (NV * NA + NV) + 2 * (NX * NA + NX);
constexpr int R16_STAGE = ((ES_QV + NA + 127) / 128) * 128;          // staged head of an edge record [A B | c | Q~ | q~ + r_y] (dompc_riccati16.h)
constexpr int R16_NEED = R16_ENABLED ? 2 * R16_STAGE : 0;
constexpr int el_max(int a, int b) { return a > b ? a : b; }
// Dense image of the model-output record (MO_COMPACT) in the wavefront's LDS region; device: the compact record of the edge
// is copied into a staging buffer next to it by the LDS-DMA path one edge ahead (eval_edge_coop), 64 lanes x 16 B per
// instruction, and scattered into the image at the top of the edge
#ifndef DOMPC_HOST_EMU
constexpr bool MO_LDS = MO_COMPACT;
#else
constexpr bool MO_LDS = false;
#endif
constexpr int MO_IMG = MO_COMPACT ? MO_SIZE : 0;
constexpr int MOC_STAGE = MO_LDS ? ((MOC_SIZE + 127) / 128) * 128 : 0;
constexpr int EL_MOS = ((EL_RT + RT_LEN + 1) / 2) * 2;                    // image (16-byte aligned)
constexpr int EL_MOC = EL_MOS + MO_IMG;                                   // staging buffer of the compact record
// forward pass, same condition: the per-edge record [G_cc^-1 | Sigma_w | r_w] and the compact model-output record of the
// NEXT edge are staged behind the step vectors while the current edge is computed; the image follows
constexpr int RF_EW = ((RF_NEED + 1) / 2) * 2;
constexpr int EW_STAGE = MO_LDS ? ((EW_SIZE + 127) / 128) * 128 : 0;
constexpr int RF_MOC = RF_EW + EW_STAGE;
constexpr int RF_IMG = RF_MOC + MOC_STAGE;
constexpr int MOH_H0 = NX + NX * NA;                                    // offset of the packed Hessian inside a point record
// DAE models: dense edge working set of eval_edge_dae (= dae::DG_SIZE, asserted in sweep())
constexpr int DAE_NEED = DENSE_EDGE ? NW * (NW + NA + 2) + (NW + NA) * (NW + NA) + (NW + NA) * (NA + 2) + 2 * (NW + NA) + 3 * NW
                                      + NX * NW + NX * NA + NX + NE * NW + NE * NA + 2 * NW + RT_LEN : 0;
// Quad sweep (round 6, dompc_quad.h): FOUR edges per wavefront, 16 lanes per edge, no dense image of the model-output record; models with
// one finite element per interval, at most 14 stage variables (x, u) and at most four nl_cons rows evaluated at (x_n, u_n).  LDS of a
// wavefront: two banks of four compact records (the next four edges are copied in by LDS-DMA while the current ones are computed), one
// slot (NX rows) of the four W | w0 matrices, the barrier vectors r_w | b of the four edges, and the transposed q~ | W'b.
#ifndef DOMPC_QUAD
#define DOMPC_QUAD 1
#endif
#ifndef DOMPC_HOST_EMU
constexpr bool QUAD_EDGE = (DOMPC_QUAD != 0) && MO_COMPACT && (DEG >= 1) && (NA + 2 <= 16) && (NE <= 4) && !RT_CUSTOM && !FREE_ROOT &&
                           (DOMPC_SHARD == 0) && (DEG * DEG * NX <= 64) && (DOMPC_DYN_NV >= NX);
#else
constexpr bool QUAD_EDGE = false;
#endif
// ... and the per-edge part of the forward pass on the same layout, the inverse G_cc^-1 formed again instead of read back (dompc_quad.h).
// The sweep then stores the inverse only for the adjoint variant of the forward pass (last barrier levels), which still reads it:
// lu_store_rule() - the barrier parameter of the sweep is at most one level above the adjoint threshold - and Prob::lu_ok.
#ifndef DOMPC_QUAD_FORWARD
#define DOMPC_QUAD_FORWARD 1
#endif
#ifndef DOMPC_ADJ_REFINE
#define DOMPC_ADJ_REFINE 1
#endif
#ifndef DOMPC_ADJ_MU
#define DOMPC_ADJ_MU 10.0
#endif
constexpr bool QUAD_FWD = QUAD_EDGE && (DOMPC_QUAD_FORWARD != 0);
// Does a sweep at barrier parameter mu store G_cc^-1 in the forward records?  Always, unless the forward pass forms it again (QUAD_FWD);
// then only if the next forward pass may be the adjoint variant (mu <= DOMPC_ADJ_MU tol, dompc_forward.h): the barrier parameter can drop by
// one level between a sweep and the solve that follows it (monotone update, refresh_mu), mu+ = min(kappa_mu mu, mu^theta_mu).  A drop by
// several levels at one iterate is caught by the driver (it repeats the sweep with Prob::soc bit 2 = "store").
DOMPC_DEV inline bool lu_store_rule(const KArgs& A, double mu, int soc) {
  if (!QUAD_FWD || (soc & 4)) return true;
  if (!DOMPC_ADJ_REFINE || (soc & 2)) return false;
  const double thr = DOMPC_ADJ_MU * A.opt.tol;
  return mu > 0.0 && mu <= fmax(thr / A.opt.kappa_mu, pow(thr, 1.0 / A.opt.theta_mu));
}

constexpr int QL_MOSZ = QUAD_EDGE ? ((4 * MO_REC + 127) / 128) * 128 : 0;     // one bank: the compact records of four consecutive edges, as they lie in memory
constexpr int QL_WS = NA + 1;                                                 // row stride of the staged W | w0 slot
constexpr int QL_WBG = ((NX * QL_WS + 1) / 2) * 2;                            // ... per edge
constexpr int QL_WB = 2 * QL_MOSZ;
constexpr int QL_VG = ((2 * NW + 7) / 8) * 8;                                 // r_w | b of one edge
constexpr int QL_RW = QL_WB + 4 * QL_WBG;
constexpr int QL_QV = QL_RW + 4 * QL_VG;                                      // q~ | W'b of one edge: 2 x 16
constexpr int qf_pad4(int n) { return n + ((4 - n % 16) + 16) % 16; }
constexpr int QF_VG = qf_pad4(2 * NW + NA + DEG * NX);                        // forward pass with four edges per wavefront: dw | dy | rhs | rr per edge (4 mod 16 doubles apart: banks)
constexpr int QF_NEED = QUAD_FWD ? QL_WB + 4 * QF_VG : 0;
constexpr int QL_NEED = QUAD_EDGE ? el_max(QL_QV + 4 * 32, QF_NEED) : 0;
// four scenario chains per wavefront in the backward Riccati pass (round 6, dompc_riccati4.h): four staged edge-record heads, four packed
// value functions, four closed-loop maps
#ifndef DOMPC_HOST_EMU
constexpr bool R4_SIZES = R16_ENABLED && (DOMPC_NE == 0) && (DOMPC_NS == 0) && (NYT <= 16);
#else
constexpr bool R4_SIZES = false;
#endif
constexpr int r4_pad4(int n) { return n + ((4 - n % 16) + 16) % 16; }
constexpr int R4_NEED = R4_SIZES ? 4 * r4_pad4(((ES_QV + NA + 31) / 32) * 32) + 4 * r4_pad4(NA * (NA + 1) / 2 + NA) + 4 * r4_pad4(NA * NA > 64 ? NA * NA : 64) : 0;
constexpr int EL_SIZE = ((el_max(el_max(el_max(el_max(el_max(EL_MOC + MOC_STAGE, RB_NEED), el_max(RF_IMG + MO_IMG, R16_NEED)), DAE_NEED), QL_NEED), R4_NEED) + 7) / 8) * 8;

// ---- dense image of a compact model-output record
// dense index (MO_PT / MO_LT / MO_MT / MO_NL layout) of compact entry k
DOMPC_DEV inline int moc_dense_index(int k) {
  constexpr int NVD = DOMPC_DYN_NV > 0 ? DOMPC_DYN_NV : 1;
  if (k < MOC_LT) return MO_PT + (k / NVD) * PT_STRIDE + DOMPC_DYN_VIDX[k % NVD];
  if (k < MOC_MT) return MO_LT + DOMPC_LT_VIDX[k - MOC_LT];
  if (k < MOC_NL) return MO_MT + DOMPC_MT_VIDX[k - MOC_MT];
  return MO_NL + DOMPC_NL_VIDX[k - MOC_NL];
}
constexpr int MOC_PL = (MOC_N + GS_C - 1) / GS_C > 0 ? (MOC_N + GS_C - 1) / GS_C : 1;     // compact entries per lane
struct MocMap { int idx[MOC_PL]; };
// this lane's scatter targets (looked up ONCE per phase: the tables live in constant memory)
DOMPC_DEV inline MocMap moc_map(int lane, int GS) {
  MocMap m;
#pragma unroll
  for (int q = 0; q < MOC_PL; ++q) {
    const int k = lane + q * GS;
    m.idx[q] = moc_dense_index(k < MOC_N ? k : 0);
  }
  return m;
}
// image <- zeros + the model's constants (once per phase and wavefront; the variable entries are overwritten per edge)
DOMPC_DEV inline void mo_image_init(ldsd* img, int lane, int GS) {
  for (int i = lane; i < MO_SIZE; i += GS) img[i] = 0.0;
#ifndef DOMPC_HOST_EMU
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#endif
  for (int i = lane; i < NCOLL * DOMPC_DYN_NC; i += GS)
    img[MO_PT + (i / (DOMPC_DYN_NC > 0 ? DOMPC_DYN_NC : 1)) * PT_STRIDE + DOMPC_DYN_CIDX[i % (DOMPC_DYN_NC > 0 ? DOMPC_DYN_NC : 1)]] =
        DOMPC_DYN_CVAL[i % (DOMPC_DYN_NC > 0 ? DOMPC_DYN_NC : 1)];
  for (int i = lane; i < DOMPC_LT_NC; i += GS) img[MO_LT + DOMPC_LT_CIDX[i]] = DOMPC_LT_CVAL[i];
  for (int i = lane; i < DOMPC_MT_NC; i += GS) img[MO_MT + DOMPC_MT_CIDX[i]] = DOMPC_MT_CVAL[i];
  for (int i = lane; i < DOMPC_NL_NC; i += GS) img[MO_NL + DOMPC_NL_CIDX[i]] = DOMPC_NL_CVAL[i];
#ifndef DOMPC_HOST_EMU
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#endif
}
// variable entries of one edge -> image.  `src`: the compact record (device: its staged copy in LDS; host: global memory)
template <class SRC>
DOMPC_DEV inline void mo_expand(ldsd* img, SRC src, const MocMap& m, int lane, int GS) {
#pragma unroll
  for (int q = 0; q < MOC_PL; ++q) {
    const int k = lane + q * GS;
    if (k < MOC_N) img[m.idx[q]] = src[k];
  }
#ifndef DOMPC_HOST_EMU
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#endif
}

DOMPC_DEV inline int point_of_slot(int sl) {
  // collocation point (i*DEG + j-1) stored in slot sl, or -1 for element-start states and xkf
  if (sl < DEG) return sl;
  const int s2 = sl - DEG, i = 1 + s2 / (DEG + 1), r = s2 % (DEG + 1);
  return (r == 0 || i >= NI) ? -1 : i * DEG + r - 1;
}

// The generated functions behind interfaces that say what the caller knows: the output record overlaps none of the inputs.  Without it the
// compiler keeps every load of an input that follows a store to the record in program order BEHIND that store, and the wait for such a
// load (s_waitcnt vmcnt is in order) is a wait for the store to reach memory.
#ifndef DOMPC_EVAL_NOALIAS
#define DOMPC_EVAL_NOALIAS 1          // 0: the generated functions as they are (A/B)
#endif
#if DOMPC_EVAL_NOALIAS
#define DOMPC_RESTRICT __restrict__
#else
#define DOMPC_RESTRICT
#endif
DOMPC_DEV inline void dyn_c_noalias(const double* DOMPC_RESTRICT xs, const double* DOMPC_RESTRICT us, const double* DOMPC_RESTRICT tvp,
                                    const double* DOMPC_RESTRICT pp, const double* DOMPC_RESTRICT lam, double* DOMPC_RESTRICT o) {
  dompc_dyn_c(xs, us, nullptr, tvp, pp, lam, o);
}
DOMPC_DEV inline void lterm_c_noalias(const double* DOMPC_RESTRICT xs, const double* DOMPC_RESTRICT us, const double* DOMPC_RESTRICT tvp,
                                      const double* DOMPC_RESTRICT pp, double* DOMPC_RESTRICT o) {
  dompc_lterm_c(xs, us, nullptr, tvp, pp, o);
}
DOMPC_DEV inline void mterm_c_noalias(const double* DOMPC_RESTRICT xs, const double* DOMPC_RESTRICT tvp, const double* DOMPC_RESTRICT pp, double* DOMPC_RESTRICT o) {
  dompc_mterm_c(xs, tvp, pp, o);
}
// Thread-parallel evaluation of the lowered model functions at the current iterate: one thread per
// (edge, function instance) - NCOLL collocation points (f, J, lambda-weighted H), stage cost,
// terminal cost (last stage), nonlinear constraints.  This is nlp_jac_g / nlp_hess_l / nlp_grad_f of
// the reference, evaluated block-wise.
DOMPC_PHASE void eval_models(const Thr& T, const Prob& Q) {
  const KArgs& A = *Q.A;
  // Work items in FUNCTION-MAJOR order: all collocation points, then all stage costs, the terminal costs of the
  // last-stage edges, the nl_cons blocks.  (Edge-major order puts every function type into every wavefront, which then
  // runs all of them one after the other with a fraction of its lanes; terminal-cost and nl_cons items of edges that have
  // none were idle slots.)
  constexpr int NPT = NPT_E;
  const int E = A.n_edges;
  const int e_last0 = E - (A.level_node_start[A.N + 1] - A.level_node_start[A.N]);      // first edge of the last stage (edges are ordered by stage)
  const int n_dyn = E * NPT, n_lt = E, n_mt = E - e_last0, n_nl = (NE > 0) ? E : 0;
  for (int it = T.tid; it < n_dyn + n_lt + n_mt + n_nl; it += T.nt) {
    int kind, e, j = 0;
    if (it < n_dyn) { kind = 0; e = it / NPT; j = it % NPT; }
    else if (it < n_dyn + n_lt) { kind = 1; e = it - n_dyn; }
    else if (it < n_dyn + n_lt + n_mt) { kind = 2; e = e_last0 + (it - n_dyn - n_lt); }
    else { kind = 3; e = it - n_dyn - n_lt - n_mt; }
    if (!mk_e(A, e)) continue;
    if constexpr (DENSE_EDGE) { dae_eval_item(Q, kind, e, j); continue; }
    const int n = A.edge_parent[e], cn = A.edge_child[e], k = A.edge_level[e];
    const double* xn = Q.x + A.node_x_off[n];
    const double* un = Q.x + A.node_u_off[n];
    const double* w = Q.x + A.edge_w_off[e];
    const double* pp = Q.P + A.p_off_p + A.edge_pidx[e] * NP;
    const double* tvp = Q.P + A.p_off_tvp + k * NTVP;
    const int row0 = A.edge_row0[e];
    double* mo = Q.MO(e);
    if constexpr (MO_COMPACT) {
      // compact record: [variable entries of point 0 | point 1 | ... | stage cost | terminal cost | nl_cons]
      if (kind == 0) {
        const int jj = j % (DEG > 0 ? DEG : 1) + 1;
        dyn_c_noalias(w + slot_of(0, jj) * NX, un, tvp, pp, Q.lam + row0 + (jj - 1) * NX, mo + j * DOMPC_DYN_NV);
      } else if (kind == 1) {
        lterm_c_noalias(xn, un, tvp, pp, mo + MOC_LT);
#if DOMPC_XTRA
        dompc_xtra_lt_c(DOMPC_XTRA_LT_ID[e], xn, un, Q.P, mo + MOC_LT);
#endif
      } else if (kind == 2) {
        if (k == A.N - 1) {
          mterm_c_noalias(Q.x + A.node_x_off[cn], Q.P + A.p_off_tvp + (k + 1) * NTVP, pp, mo + MOC_MT);
#if DOMPC_XTRA
          dompc_xtra_mt_c(DOMPC_XTRA_MT_ID[e], Q.x + A.node_x_off[cn], Q.P, mo + MOC_MT);
#endif
        }
      } else if (NE > 0) {
        double yds[NE1];      // (scaled rows sg d(x): the Hessian sum_i lambda_i sg_i hess d_i)
        for (int i = 0; i < NE; ++i) yds[i] = Q.lam[row0 + NW + NX + i] * Q.sgn[e * NE1 + i];
        dompc_nlcons_c(xn, un, nullptr, tvp, pp, yds, mo + MOC_NL);
#if DOMPC_XROW
        for (int r = 0; r < DOMPC_XROW_SLOTS; ++r) dompc_xrow_c(DOMPC_XROW_ID[e * DOMPC_XROW_SLOTS + r], xn, un, Q.P, yds, mo + MOC_NL);
#endif
      }
    } else if (kind == 0) {
      double* pt = mo + MO_PT + j * PT_STRIDE;
      if (M == 0) {
        dompc_dyn(xn, un, nullptr, tvp, pp, Q.lam + row0 + NW, pt, pt + NX, pt + NX + NX * NA);
      } else {
        const int i = j / DEG, jj = j % DEG + 1;
        dompc_dyn(w + slot_of(i, jj) * NX, un, nullptr, tvp, pp, Q.lam + row0 + i * (DEG + 1) * NX + (jj - 1) * NX,
                  pt, pt + NX, pt + NX + NX * NA);
      }
    } else if (kind == 1) {
      lterm_e(Q, e, xn, un, nullptr, tvp, pp, mo + MO_LT, mo + MO_LT + 1, mo + MO_LT + 1 + NA);
    } else if (kind == 2) {
      if (k == A.N - 1)
        mterm_e(Q, e, Q.x + A.node_x_off[cn], Q.P + A.p_off_tvp + (k + 1) * NTVP, pp, mo + MO_MT, mo + MO_MT + 1,
                    mo + MO_MT + 1 + NX);
    } else if (NE > 0) {
      double yds[NE1];
      for (int i = 0; i < NE; ++i) yds[i] = Q.lam[row0 + NW + NX + i] * Q.sgn[e * NE1 + i];
      nlcons_e(Q, e, xn, un, nullptr, tvp, pp, yds, mo + MO_NL, mo + MO_NL + NE, mo + MO_NL + NE + NE * NA);
    }
  }
}

#ifndef DOMPC_HOST_EMU
// 16x16 FP64 tiles in the accumulator layout of v_mfma_f64_16x16x4_f64 (lane l holds M[(l >> 4) + 4 r][l & 15], r = 0..3):
// such a tile is directly the B operand of k-block r and, as A operand, the TRANSPOSED matrix, so
// tile_mul(X, Y) = X' Y is KB back-to-back MFMAs on registers (see dompc_riccati16.h).
typedef double d4 __attribute__((ext_vector_type(4)));
template <int KB>
__device__ inline d4 tile_mul(const d4& At, const d4& B) {      // At' * B over the first 4*KB rows of both
  d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(At[kb], B[kb], acc, 0, 0, 0);
  return acc;
}
#endif

// value of `v` in lane `src` (wave-uniform, here a compile-time constant) for every lane: two v_readlane_b32, the
// result lives in SGPRs.  Host emulation (one lane): the value itself.
DOMPC_DEV inline double lane_bcast(double v, int src) {
#ifndef DOMPC_HOST_EMU
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
#else
  (void)src;
  return v;
#endif
}

