// dompc_riccati16.h - tree Riccati recursion on the FP64 matrix cores with register-resident tiles (gfx950 only).
//
// The node quadratic of the reference NLP lives over z = (x, u_prev, u [, eps]) - 16 entries for industrial_poly - and
// every matrix of the backward recursion (P_c, F = [[A|0|B],[0|0|I]], Q, the gains) fits ONE 16x16 tile of
// v_mfma_f64_16x16x4_f64.  A tile is kept in the ACCUMULATOR layout of that instruction,
//     lane l holds  M[(l >> 4) + 4 r][l & 15],  r = 0..3        (4 doubles per lane),
// and the operand layouts of the instruction make such a tile directly usable
//     * as the B operand of k-block r  (B[4r + (l>>4)][l&15]  = element r of the lane), and
//     * as the A operand of k-block r  for the TRANSPOSED matrix (A[l&15][4r + (l>>4)] = M'[..]),
// so  tmul(X, Y) = X' Y  is four back-to-back MFMAs on registers: products chain without LDS, barriers or data
// movement (P_c is symmetric, F' P_c F = tmul(F, tmul(P_c, F)), L' Q L = tmul(L, tmul(Q, L)), ...).
// Vectors ride INSIDE the matrix tiles (homogeneous coordinates): on gfx950 an FP64 MFMA has the throughput of the FP64 vector
// ALU (64 cycles per 16x16x4), so a matrix-vector product in a tile of its own costs as much as the matrix-matrix product next
// to it.  The value function of a node is ONE bordered symmetric tile  V = [[P, p], [p', *]]  (p in column NA and row NA - P is
// NA x NA with NA <= 15); the child's map F = [A 0 B; 0 0 I] has structurally zero columns at the u_prev entries of z, so
// column NX carries [c; 1 at row NA]:  F1' V F1  then holds F' P F outside row / column NX and F'(P c + p) in them; the gains
// Lc = [I; K] and the closed-loop maps Acl = F Lc have free columns >= NA, column NA carries l0 = (0; kv) resp. [F l0 + c; 1].
// 25 instead of 50 MFMAs per node and child.
// This replaces the generic LDS-staged riccati_node() (dompc_riccati.h; still used by the host emulation, by models
// with more than 16 node variables, more than 4 decision variables per node or more than 4 nl_cons rows, and by the
// tree-sharding build) - the algebra is the same:
//     Q_tot = Q_own + sum_c F_c' P_c F_c ,  K = -Q_vv^-1 Q_vx ,
//     P = Lc' Q_own Lc + sum_c Acl_c' P_c Acl_c   (closed-loop "Joseph" form, Lc = [I;K], Acl = F Lc).
#pragma once

namespace dompc {
namespace r16 {
constexpr bool ENABLED = R16_ENABLED;      // (NYT <= 16, at most 4 nl_cons rows, not the tree-sharding build)

#ifndef DOMPC_HOST_EMU
constexpr int KB_A = (NA + 3) / 4, KB_Y = (NYT + 3) / 4;
constexpr int KB_H = (NA + 4) / 4;                 // k-blocks that cover the NA rows of a child's state AND the homogeneous row NA
constexpr int HR = NA / 4, HG = NA % 4;            // register / lane group of tile row NA
static_assert(!R16_ENABLED || NA <= 15, "the homogeneous row / column needs NA <= 15");
template <int KB>
__device__ inline d4 tmul(const d4& At, const d4& B) { return tile_mul<KB>(At, B); }      // At' * B (dompc_kernel.h)
__device__ inline double rl(double v, int src) { return lane_bcast(v, src); }

// index of z-entry i inside y = (x_n, u_n) of the condensed edge blocks, or -1 (u_prev, eps)
__device__ inline int yz(int i) { return (i < NX) ? i : ((i >= NA && i < NA + NU) ? NX + (i - NA) : -1); }

// column-layout vector (lane l holds v[l & 15]) -> column `col` of a tile (rows < nrow)
__device__ inline d4 col_to_tile(double vc, int lane, int col, int nrow) {
  const int g = lane >> 4, j = lane & 15;
  d4 t;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double s = __shfl(vc, g + 4 * r);
    t[r] = (j == col && g + 4 * r < nrow) ? s : 0.0;
  }
  return t;
}

typedef d4 Val;                // value function of a node: bordered tile [[P, p], [p', *]] (p in column NA and row NA)

// F tile, c column and rank-update operand of edge e
__device__ inline void load_edge(const Prob& Q, int e, int lane, d4& F, d4& cvr, double& fu) {
  const int g = lane >> 4, j = lane & 15;
  const double* S_ = Q.ES(e);
  const int yj = yz(j);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = g + 4 * r;
    double fv = 0.0;
    if (i < NX) { if (yj >= 0 && j < NYT) fv = S_[ES_AB + i * NA + yj]; }
    else if (i < NA) fv = (j == NA + (i - NX)) ? 1.0 : 0.0;
    F[r] = fv;
    cvr[r] = (i < NX) ? S_[ES_CV + i] : 0.0;
  }
  double v = 0.0;
  if (g < NU) { if (j < NX) v = S_[ES_AB + j * NA + NX + g]; else if (j < NA) v = (g == j - NX) ? 1.0 : 0.0; }
  fu = v;
}

// condensed Hessian block of edge e (packed upper triangle) scattered into the z x z tile
__device__ inline d4 load_qt(const Prob& Q, int e, int lane) {
  const int g = lane >> 4, j = lane & 15;
  const double* S_ = Q.ES(e);
  const int yj = yz(j);
  d4 t;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = g + 4 * r;
    const int yi = yz(i);
    const bool valid = i < NYT && j < NYT && yi >= 0 && yj >= 0;
    t[r] = valid ? S_[ES_QT + symi(yi, yj, NA)] : 0.0;
  }
  return t;
}

__device__ inline Val load_val(const Prob& Q, int n, int lane) {
  const int g = lane >> 4, j = lane & 15;
  const double* Nd = Q.ND(n);
  Val V;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = g + 4 * r;
    double v = (i < NA && j < NA) ? Nd[ND_P + i * NA + j] : 0.0;
    if (i < NA && j == NA) v = Nd[ND_PV + i];
    if (i == NA && j < NA) v = Nd[ND_PV + j];
    V[r] = v;
  }
  return V;
}
__device__ inline void store_val(const Prob& Q, int n, const Val& V, int lane) {
  const int g = lane >> 4, j = lane & 15;
  double* Nd = Q.ND(n);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = g + 4 * r;
    if (i < NA && j < NA) Nd[ND_P + i * NA + j] = V[r];
    if (i < NA && j == NA) Nd[ND_PV + i] = V[r];
  }
}

// leaf: P = sf*omega*Hm + Sigma_x + delta, p = sf*omega*gm - nu_in + barrier gradient   (x part only)
__device__ inline Val leaf(const Prob& Q, int n, double mu, double delta, int lane) {
  const KArgs& A = *Q.A;
  const int g = lane >> 4, j = lane & 15;
  const int ie = A.node_in_edge[n];
  const double* S_ = Q.ES(ie);
  const int xo = A.node_x_off[n];
  const int jj = j < NX ? j : 0;
  const double xv = Q.x[xo + jj], lo = Q.lb[xo + jj], hi = Q.ub[xo + jj];
  const double dg = sigma_of(xv, lo, hi, Q.zl[xo + jj], Q.zu[xo + jj]) + delta;
  const double gv = (j < NX) ? S_[ES_MG + jj] - Q.lam[A.edge_row0[ie] + NW + jj] + bar_grad(xv, lo, hi, mu, !(Q.soc & 2)) : 0.0;
  Val V = col_to_tile(gv, lane, NA, NX);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = g + 4 * r;
    double v = (i < NX && j < NX) ? S_[ES_MH + i * NX + j] : 0.0;
    if (i == j && j < NX) v += dg;
    if (i == NA && j < NX) v = gv;
    V[r] += v;
  }
  return V;
}

// Operands of one node update that come from global memory, requested ONE NODE AHEAD while a wavefront walks its
// scenario chain (they arrive during the ~50 matrix-core instructions of the child's update instead of costing a
// memory round trip at the start of every node):
//   * the node's own variable data (column layout: lane l, z-entry l & 15) - NodeIn, 7 registers;
//   * the head of the first child edge's record [A B | c | Q~ packed | q~ + r_y] (ES_STAGE doubles, contiguous) - copied
//     asynchronously into the wavefront's LDS region by the LDS-DMA path (global_load_lds_dwordx4: no staging
//     registers), two buffers alternating between consecutive nodes.
constexpr int ES_STAGE = R16_STAGE;                 // doubles: whole 64 lanes x 16 B pieces covering [0, ES_QV + NA)
static_assert(ES_QV + NA <= ES_STAGE && ES_QT + NA_T <= ES_QV && ES_CV + NX <= ES_QT,
              "the staged head of the edge record must contain A, B, c, Q~, q~ + r_y");
// (a piece may run past the end of a short record into the following records of the same workspace slot - never used;
//  ws_layout() keeps 128 doubles of slack behind the last array)
static_assert(!R16_ENABLED || 2 * ES_STAGE <= EL_SIZE, "two staging buffers must fit the wavefront's LDS region");
struct NodeIn { double xv, lo, hi, zlo, zhi, nu, upv; };
__device__ inline void stage_edge(const Prob& Q, int e, int lane, ldsd* dst) {
  const double* S_ = Q.ES(e);
#pragma unroll
  for (int q = 0; q < ES_STAGE / 128; ++q)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(S_ + 128 * q + 2 * lane),
                                     (__attribute__((address_space(3))) void*)(dst + 128 * q), 16, 0, 0);
}
__device__ inline void staged_ready() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ inline void load_node(const Prob& Q, int n, int lane, NodeIn& R) {
  const KArgs& A = *Q.A;
  const int j = lane & 15;
  const int xo = A.node_x_off[n], uo = A.node_u_off[n];
  const int ie = A.node_in_edge[n], pn = A.node_parent[n];
  const bool is_up = (j >= NX && j < NA);
  const int jj = j < NYT ? j : 0;
  const int eo = NS > 0 ? A.node_eps_off[n] : 0;
  const bool is_eps = NS > 0 && jj >= NA + NU;
  const int gi = (jj < NX) ? xo + jj : (is_up ? uo + (jj - NX) : (is_eps ? eo + (jj - NA - NU) : uo + (jj - NA)));
  R.xv = Q.x[gi]; R.lo = Q.lb[gi]; R.hi = Q.ub[gi]; R.zlo = Q.zl[gi]; R.zhi = Q.zu[gi];
  const int iu = is_up ? jj - NX : ((jj >= NA && !is_eps) ? jj - NA : 0);
  R.upv = (jj >= NX && !is_eps) ? (pn >= 0 ? Q.x[A.node_u_off[pn] + iu] : Q.P[A.p_off_uprev + iu] / tab_sel(DOMPC_SU, iu)) : 0.0;
  R.nu = (jj < NX) ? ((ie >= 0) ? Q.lam[A.edge_row0[ie] + NW + jj] : Q.lam[jj]) : 0.0;
}
// tiles of the staged first child edge (LDS reads)
__device__ inline void staged_tiles(const ldsd* Ls, int lane, d4& qt, d4& F, d4& cvr, double& fu, double& qv) {
  const int g = lane >> 4, j = lane & 15;
  const int yj = yz(j);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = g + 4 * r;
    const int yi = yz(i);
    const bool valid = i < NYT && j < NYT && yi >= 0 && yj >= 0;
    const double q_ = Ls[ES_QT + (valid ? symi(yi, yj, NA) : 0)];
    qt[r] = valid ? q_ : 0.0;
    const double ab = Ls[ES_AB + ((i < NX && yj >= 0) ? i * NA + yj : 0)];
    double fv = (i < NX && yj >= 0 && j < NYT) ? ab : 0.0;
    if (i >= NX && i < NA) fv = (j == NA + (i - NX)) ? 1.0 : 0.0;
    F[r] = fv;
    const double cv = Ls[ES_CV + (i < NX ? i : 0)];
    cvr[r] = (i < NX) ? cv : 0.0;
  }
  {
    const double ab = Ls[ES_AB + ((j < NX && g < NU) ? j * NA + NX + g : 0)];
    double v = (g < NU && j < NX) ? ab : 0.0;
    if (g < NU && j >= NX && j < NA) v = (g == j - NX) ? 1.0 : 0.0;
    fu = v;
  }
  const int yjj = (j < NYT) ? yj : -1;
  qv = (yjj >= 0) ? Ls[ES_QV + (yjj >= 0 ? yjj : 0)] : 0.0;      // (q~ + r_y)
}

// One node update.  `first`: value function of the first child when it is already in registers (chain walk), else
// null and every child's value function is read from its node record.  Returns 1 if Q_vv is not positive definite.
__device__ inline int node(const Prob& Q, int n, double mu, double delta, int lane, const NodeIn& R, const ldsd* Ls, const Val* first, Val& out) {
  const KArgs& A = *Q.A;
  const int g = lane >> 4, j = lane & 15;
  const int cs = A.node_child_start[n], cc = A.node_child_count[n];
  const double rw = node_rweight(Q, n);
  const double rwh = (Q.soc & 2) ? 0.0 : rw;          // weight of the rterm HESSIAN (Prob::soc bit 1: least-squares multiplier solve)
  long long pc0 = prof_clock();
#if DOMPC_PROFILE
#define R16_PN(i) if (threadIdx.x == 0) { const long long pc1 = prof_clock(); lds_prof[i] += pc1 - pc0; pc0 = pc1; }
#else
#define R16_PN(i)
#endif
  d4 qt_s, F, cvr;
  double fu, qv_s;
  staged_tiles(Ls, lane, qt_s, F, cvr, fu, qv_s);
  DOMPC_PRIO_UP();
  R16_PN(12)
  // ---- own quadratic: per-variable terms in column layout (lane: z-entry j)
  double dg = 0.0, gv = 0.0;
  {
    const int ie = A.node_in_edge[n];
    const bool is_up = (j >= NX && j < NA);
    const int jj = j < NYT ? j : 0;
    const bool is_eps = NS > 0 && jj >= NA + NU;
    const int iu = is_up ? jj - NX : ((jj >= NA && !is_eps) ? jj - NA : 0);
    const double xv = R.xv, lo = R.lo, hi = R.hi, upv = R.upv;
    if (is_up) {
      dg = 2.0 * rwh * tab_sel(DOMPC_RTERM, iu);
      gv = -2.0 * rw * tab_sel(DOMPC_RTERM, iu) * (xv - upv);
    } else {
      dg = sigma_of(xv, lo, hi, R.zlo, R.zhi) + delta;
      gv = bar_grad(xv, lo, hi, mu, !(Q.soc & 2));
      if (jj < NX) {
        gv += (ie >= 0) ? -R.nu : R.nu;
      } else if (is_eps) {
        if constexpr (NS > 0) gv += cc * Q.sf * tab_sel(DOMPC_EPS_PEN, jj - NA - NU);      // slack penalty (one term per outgoing edge)
      } else {
        dg += 2.0 * rwh * tab_sel(DOMPC_RTERM, iu);
        gv += 2.0 * rw * tab_sel(DOMPC_RTERM, iu) * (xv - upv);
      }
    }
    const int yjj = yz(jj);
    gv += qv_s;
    if (yjj >= 0)
      for (int c = 1; c < cc; ++c) gv += Q.ES(cs + c)[ES_QV + yjj];
    if constexpr (NE > 0) {
      // nl_cons rows of the child edges, condensed through their slacks: gradient share  J~'((Sigma_s + delta) r_d + r_s)
      // with J~ = [J_d over (x, u) | -1 at the slack variable of the row]  (same algebra as riccati_node, dompc_riccati.h)
      for (int c = 0; c < cc; ++c) {
        const int e = cs + c;
        const double* S_ = Q.ES(e);
#pragma unroll
        for (int q = 0; q < NE; ++q) {
          const double sg = S_[ES_SIGS + q] + delta;
          const bool slack_here = NS > 0 && jj >= NA + NU && nl_slack(q) == jj - NA - NU;
          const double jc = (yjj >= 0) ? Q.EW(e, EW_JD + q * NA + yjj) : (slack_here ? -Q.sgn[e * NE1 + q] : 0.0);     // (scaled row sg (d - eps))
          gv += jc * (sg * S_[ES_RDN + q] + S_[ES_RSN + q]);
          if (slack_here) gv += jc * Q.lam[A.edge_row0[e] + NW + NX + q];
        }
      }
    }
    if (j >= NYT) { dg = 0.0; gv = 0.0; }
  }
  R16_PN(13)
  d4 QO = qt_s;
  for (int c = 1; c < cc; ++c) QO += load_qt(Q, cs + c, lane);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = g + 4 * r;
    if (i == j) QO[r] += dg;
    else if ((i >= NX && i < NA && j == i + NU) || (j >= NX && j < NA && i == j + NU))
      QO[r] -= 2.0 * rwh * tab_sel(DOMPC_RTERM, (i < j ? i : j) - NX);
  }
  R16_PN(14)
  if constexpr (NE > 0) {
    // ... and the Hessian share  sum_q (Sigma_s + delta) J~_q' J~_q  as rank-one updates of the own tile
    const int jz = j < NYT ? j : 0, yj = yz(jz);
    for (int c = 0; c < cc; ++c) {
      const int e = cs + c;
      const double* S_ = Q.ES(e);
#pragma unroll
      for (int q = 0; q < NE; ++q) {
        const double sg = S_[ES_SIGS + q] + delta;
        const double jc = (j >= NYT) ? 0.0 : ((yj >= 0) ? Q.EW(e, EW_JD + q * NA + yj)
                                                        : ((NS > 0 && jz >= NA + NU && nl_slack(q) == jz - NA - NU) ? -Q.sgn[e * NE1 + q] : 0.0));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = g + 4 * r, iz = i < NYT ? i : 0, yi = yz(iz);
          const double jr = (i >= NYT) ? 0.0 : ((yi >= 0) ? Q.EW(e, EW_JD + q * NA + yi)
                                                          : ((NS > 0 && iz >= NA + NU && nl_slack(q) == iz - NA - NU) ? -Q.sgn[e * NE1 + q] : 0.0));
          QO[r] += sg * jr * jc;
        }
      }
    }
  }
  const d4 qo0 = col_to_tile(gv, lane, NA, 16);            // own gradient as column NA of a tile (own part below)
  R16_PN(8)
  // ---- children, pass 1: Q_tot = Q_own + sum F' P_c F ,  q_tot = q_own + sum F'(P_c f + p_c).  F1 = F with [c; 1 at the
  //      homogeneous row] in its zero column NX: F1' V F1 = F' P F outside row / column NX, the vector in them
  d4 QT = QO;
  double qvs[NV];                      // decision-variable entries of q_tot (uniform)
#pragma unroll
  for (int u = 0; u < NV; ++u) qvs[u] = rl(gv, NA + u);
  Val Vc;
  for (int c = 0; c < cc; ++c) {
    if (c > 0) load_edge(Q, cs + c, lane, F, cvr, fu);
    Vc = (c == 0 && first) ? *first : load_val(Q, A.edge_child[cs + c], lane);
    d4 F1;
#pragma unroll
    for (int r = 0; r < 4; ++r) F1[r] = (j == NX) ? ((g + 4 * r == NA) ? 1.0 : cvr[r]) : F[r];
    const d4 Tm = tmul<KB_H>(Vc, F1);
    const d4 Pr = tmul<KB_H>(F1, Tm);
#pragma unroll
    for (int r = 0; r < 4; ++r) QT[r] += (j == NX || g + 4 * r == NX) ? 0.0 : Pr[r];
#pragma unroll
    for (int u = 0; u < NV; ++u) qvs[u] += rl(Pr[(NA + u) / 4], 16 * ((NA + u) % 4) + NX);
  }
  R16_PN(9)
  // ---- Cholesky of Q_vv (uniform arithmetic on values read with v_readlane), gains for this lane's column
  double L[NV * NV], kv[NV], Kj[NV];
  int bad = 0;
  {
    double qvv[NV * NV], qv[NV], qx[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int i = NA + u;
#pragma unroll
      for (int w = 0; w <= u; ++w) qvv[u * NV + w] = rl(QT[i / 4], 16 * (i % 4) + NA + w);
      qv[u] = qvs[u];
      qx[u] = __shfl(QT[i / 4], 16 * (i % 4) + j);
    }
    double Li[NV];                       // reciprocals of the Cholesky diagonal: every division below is a multiplication
#pragma unroll
    for (int u = 0; u < NV; ++u)
#pragma unroll
      for (int w = 0; w <= u; ++w) {
        double t = qvv[u * NV + w];
#pragma unroll
        for (int q = 0; q < w; ++q) t -= L[u * NV + q] * L[w * NV + q];
        if (u == w) {
          if (!(t > 0.0)) { bad = 1; t = 1.0; }
          L[u * NV + u] = sqrt(t);
          Li[u] = fast_rcp(L[u * NV + u]);
        } else {
          L[u * NV + w] = t * Li[w];
        }
      }
    auto solve = [&](double* y) {        // y <- Q_vv^-1 y
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        double t = y[u];
#pragma unroll
        for (int q = 0; q < u; ++q) t -= L[u * NV + q] * y[q];
        y[u] = t * Li[u];
      }
#pragma unroll
      for (int u = NV - 1; u >= 0; --u) {
        double t = y[u];
#pragma unroll
        for (int q = u + 1; q < NV; ++q) t -= L[q * NV + u] * y[q];
        y[u] = t * Li[u];
      }
    };
    solve(qv);
    solve(qx);
#pragma unroll
    for (int u = 0; u < NV; ++u) { kv[u] = -qv[u]; Kj[u] = (j < NA) ? -qx[u] : 0.0; }
  }
  R16_PN(10)
  // ---- Lc = [I;K | l0 = (0;kv) in column NA] as a tile; operand of the rank-NV updates
  d4 Lc;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = g + 4 * r;
    double v = (k < NA) ? ((k == j) ? 1.0 : 0.0) : 0.0;
#pragma unroll
    for (int u = 0; u < NV; ++u)
      if (k == NA + u) v = (j == NA) ? kv[u] : Kj[u];
    Lc[r] = v;
  }
  double bK = 0.0;
#pragma unroll
  for (int u = 0; u < NV; ++u)
    if (g == u) bK = (j == NA) ? kv[u] : Kj[u];
  // ---- own part of the value function: Lc' Q_own Lc, Lc'(q_own + Q_own l0) in column NA; row NA completed by Lc' q_own
  //      (z has no homogeneous entry - all 16 are variables -, so q_own enters the column by an addition and the row here)
  {
    const d4 U = tmul<KB_Y>(QO, Lc) + qo0;
    out = tmul<KB_Y>(Lc, U);
    double t = gv;
#pragma unroll
    for (int u = 0; u < NV; ++u) t += Kj[u] * rl(gv, NA + u);
    if (g == HG && j < NA) out[HR] += t;
  }
  R16_PN(11)
  // ---- children, pass 2: closed-loop maps Acl = F Lc, ccl = F l0 + f ; P += Acl' P_c Acl
  for (int c = cc - 1; c >= 0; --c) {
    if (c != cc - 1) {                       // (the last child of pass 1 is still in registers)
      load_edge(Q, cs + c, lane, F, cvr, fu);
      Vc = (c == 0 && first) ? *first : load_val(Q, A.edge_child[cs + c], lane);
    }
    d4 Fy;                                   // [F over (x, u_prev) | c ; homogeneous 1]  ->  Acl = [F Lc | F l0 + c ; 1]
#pragma unroll
    for (int r = 0; r < 4; ++r) Fy[r] = (j < NA) ? F[r] : ((j == NA) ? ((g + 4 * r == NA) ? 1.0 : cvr[r]) : 0.0);
    const d4 Acl = __builtin_amdgcn_mfma_f64_16x16x4f64(fu, bK, Fy, 0, 0, 0);
    const d4 T2 = tmul<KB_H>(Vc, Acl);
    out += tmul<KB_H>(Acl, T2);
  }
  // All global stores of the node at its very end, behind one wait for the operands that were requested for the NEXT
  // node (LDS-DMA copy, register prefetch): on gfx9 stores count in vmcnt like loads, so a wait for data issued after
  // a store also waits for the store's ~2 us round trip - waiting first and storing afterwards keeps the stores of
  // this node in flight during the whole update of the next one.
  DOMPC_PRIO_DOWN();
  R16_PN(24)
  staged_ready();
  R16_PN(25)
  double* Nd = Q.ND(n);
  if (g == 0 && j < NA)
#pragma unroll
    for (int u = 0; u < NV; ++u) Nd[ND_K + u * NA + j] = Kj[u];
  if (lane == 0)
#pragma unroll
    for (int u = 0; u < NV; ++u) Nd[ND_KV + u] = kv[u];
  store_val(Q, n, out, lane);
  R16_PN(26)
#undef R16_PN
  return bad;
}

}  // namespace r16
}  // namespace dompc
#include "dompc_riccati4.h"      // four scenario chains per wavefront (uses NodeIn / load_node above)
namespace dompc {
namespace r16 {
// Backward recursion of one problem (all wavefronts of the problem take part; same protocol as riccati_backward).
__device__ inline int backward(const Thr& T, const Prob& Q, double mu, double delta) {
  const KArgs& A = *Q.A;
  const int ng = T.nt / 64, gid = group_index(T.tid, 64), lane = T.tid % 64;
  ldsd* Ld = T.edge_lds + (int64_t)(T.ltid / 64) * EL_SIZE;      // this wavefront's LDS region: two staging buffers
  const int FSET = T.flag_begin(0);
  const int cl = A.chain_level < A.N ? A.chain_level : A.N;
  if constexpr (r4::ENABLED) {
    if (r4::chains(T, Q, mu, delta, cl)) T.fset(0, FSET);
    T.sync();
    if ((T.fget(0) == FSET)) return 1;
  } else {
    const int S = A.level_node_start[A.N + 1] - A.level_node_start[A.N];
    for (int s_ = gid; s_ < S; s_ += ng) {
      NodeIn in;
      int buf = 0;
      if (A.N - 1 >= cl) {
        const int n1 = A.level_node_start[A.N - 1] + s_;
        load_node(Q, n1, lane, in);
        stage_edge(Q, A.node_child_start[n1], lane, Ld);
      }
      Val V = leaf(Q, A.level_node_start[A.N] + s_, mu, delta, lane);
      store_val(Q, A.level_node_start[A.N] + s_, V, lane);
      staged_ready();              // the staged record of the first node has landed (later ones: waited for inside node())
      for (int k = A.N - 1; k >= cl; --k) {
        NodeIn nx;                 // the parent's operands: in flight while this node is updated
#if DOMPC_PROFILE
        const long long pcl = prof_clock();
#endif
        if (k > cl) {
          const int np = A.level_node_start[k - 1] + s_;
          load_node(Q, np, lane, nx);
          stage_edge(Q, A.node_child_start[np], lane, Ld + (buf ^ 1) * ES_STAGE);
        }
#if DOMPC_PROFILE
        if (threadIdx.x == 0) lds_prof[27] += prof_clock() - pcl;
#endif
        Val Vn;
        if (node(Q, A.level_node_start[k] + s_, mu, delta, lane, in, Ld + buf * ES_STAGE, &V, Vn)) { T.fset(0, FSET); break; }
        V = Vn;
        if (k > cl) in = nx;
        buf ^= 1;
      }
    }
    T.sync();
    if ((T.fget(0) == FSET)) return 1;
  }
  for (int k = cl - 1; k >= 0; --k) {
    const int n0 = A.level_node_start[k], n1 = A.level_node_start[k + 1];
    for (int n = n0 + gid; n < n1; n += ng) {
      NodeIn in;
      load_node(Q, n, lane, in);
      stage_edge(Q, A.node_child_start[n], lane, Ld);
      staged_ready();
      Val Vn;
      if (node(Q, n, mu, delta, lane, in, Ld, nullptr, Vn)) T.fset(0, FSET);
    }
    T.sync();
    if ((T.fget(0) == FSET)) return 1;
  }
  return 0;
}
#endif  // !DOMPC_HOST_EMU

}  // namespace r16
}  // namespace dompc
