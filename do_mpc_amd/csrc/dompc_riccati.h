// dompc_riccati.h - structured interior-point solver, part of dompc_kernel.h (included there, inside namespace dompc, in this order:
// dompc_edge.h, dompc_factor.h, dompc_node.h, dompc_riccati.h, dompc_forward.h, dompc_sweep.h, dompc_phases.h, dompc_driver.h).
// Contents: tree Riccati recursion: node update, cut parents of a sharded tree, backward pass (pulls in dompc_riccati16.h, the register-resident matrix-core recursion, at file scope: it closes and re-opens namespace dompc around that include).
// Sizes, record layouts, the thread context `Thr`, reductions and the small dense products are in dompc_kernel.h.

// ================================================================================================
// Tree Riccati recursion.  Value function of node n over its augmented state (x_n, u_prev_n):
//   V_n(d) = 1/2 d'P_n d + p_n'd   (Newton form: p built from dual residuals).
// Children are summed at branching nodes (non-anticipativity = shared variables, _mpc.py:1212-1216).
DOMPC_DEV inline int ycol(int yj) { return yj < NX ? yj : NA + (yj - NX); }

namespace rb {
// LDS working set of one node update (offsets in doubles inside the group's region)
constexpr int RB_QO = 0, RB_QOV = RB_QO + NYT * NYT;          // own quadratic (x, u_prev, u, eps) + gradient
constexpr int RB_QF = RB_QOV + NYT, RB_QFV = RB_QF + NYT * NYT; // own + children's value functions
constexpr int RB_PC = RB_QFV + NYT, RB_PCV = RB_PC + NA * NA;  // child P_c, p_c
constexpr int RB_AT = RB_PCV + NA, RB_CT = RB_AT + NA * NA;    // Atilde over y=(x_n,u_n): [[A|B],[0|I]] (NA x NA), ctilde
constexpr int RB_TP = RB_CT + NA, RB_TV = RB_TP + NA * NA;     // P_c Atilde / P_c Acl, and the vector twins
constexpr int RB_K = RB_TV + NA, RB_KV = RB_K + NV * NA;
constexpr int RB_ACL = RB_KV + NV, RB_CCL = RB_ACL + NA * NA;  // closed-loop map Atilde [I;K] (also: scratch for Atilde' TP)
constexpr int RB_PN = RB_CCL + NA, RB_PNV = RB_PN + NA * NA;   // result P_n, p_n
constexpr int RB_NL = RB_PNV + NA;                             // staged nl_cons data of one child edge
constexpr int RB_SIZE = RB_NL + NE * (NA + 4);
}  // namespace rb

// index of entry i of (x, u_prev, u, eps) inside y = (x_n, u_n), or -1
DOMPC_DEV inline int yidx(int i) { return (i < NX) ? i : ((i >= NA && i < NA + NU) ? NX + (i - NA) : -1); }

// Global operands of a node update that belong to the node itself and to its FIRST child edge, loaded into
// registers ahead of time: while a group walks its scenario chain upwards, the loads of the parent are in
// flight during the update of the child (the update used to spend ~40 % of its time waiting for exactly
// these loads).  Raw values only - anything computed from them here would stall the issuing wavefront.
constexpr int RN_IPL = (NYT * NYT + GS_C - 1) / GS_C;
constexpr int RN_VPL = (NYT + GS_C - 1) / GS_C;
constexpr int RN_NE1 = NE > 0 ? NE : 1;
constexpr int RN_NLN = NE * (NA + 4);          // nl_cons data of a child edge: [JD (NE x NA) | SIGS | RDN | RSN | y_d]
constexpr int RN_NLP = NE > 0 ? (RN_NLN + GS_C - 1) / GS_C : 1;
constexpr int RN_ABN = NX * (NA + 1);          // [A | B | c] of a child edge
constexpr int RN_ABP = (RN_ABN + GS_C - 1) / GS_C;
// (An inertia correction that the last sweep has not folded into the condensed blocks - Q~(delta) = Q~ + delta W'W - is
//  handled by REPEATING the sweep with Prob::dsw = delta (solve_problem): W is not kept beyond the sweep any more.)
struct NodePre {
  double qt[RN_IPL];
  double pv[RN_VPL][10];                       // x, lb, ub, zl, zu, nu_in, u_prev, -, q~ + r_y, -
  double nl[RN_NLP];
  double ab[RN_ABP];
};

DOMPC_DEV inline double node_nl_load(const Prob& Q, int e, int it) {
  if (it < NE * NA) return Q.EW(e, EW_JD + it);
  const int j = it - NE * NA, kind = j / RN_NE1, q = j % RN_NE1;
  const double* S_ = Q.ES(e);
  return kind == 0 ? S_[ES_SIGS + q] : kind == 1 ? S_[ES_RDN + q] : kind == 2 ? S_[ES_RSN + q]
                   : Q.lam[Q.A->edge_row0[e] + NW + NX + q];
}

DOMPC_DEV inline void node_prefetch(const Prob& Q, int n, double delta, int lane, int GS, NodePre& R) {
  const KArgs& A = *Q.A;
  const int e = A.node_child_start[n];
  const double* S_ = Q.ES(e);
  const int xo = A.node_x_off[n], uo = A.node_u_off[n];
  const int eo = NS > 0 ? A.node_eps_off[n] : -1;
  const int ie = A.node_in_edge[n], pn = A.node_parent[n];
  (void)delta;
#pragma unroll
  for (int q = 0; q < RN_IPL; ++q) {
    const int it = lane + q * GS;
    const int itc = it < NYT * NYT ? it : 0;
    const int yi = yidx(itc / NYT), yj = yidx(itc % NYT);
    const int idx = (yi >= 0 && yj >= 0) ? symi(yi, yj, NA) : 0;
    R.qt[q] = S_[ES_QT + idx];
  }
#pragma unroll
  for (int v = 0; v < RN_VPL; ++v) {
    const int i0 = lane + v * GS;
    const int i = i0 < NYT ? i0 : 0;
    const int yi = yidx(i);
    const bool is_up = (i >= NX && i < NA);
    const int g = (i < NX) ? xo + i : (is_up ? uo + (i - NX) : (i < NA + NU ? uo + (i - NA) : eo + (i - NA - NU)));
    R.pv[v][0] = Q.x[g];
    R.pv[v][1] = Q.lb[g];
    R.pv[v][2] = Q.ub[g];
    R.pv[v][3] = Q.zl[g];
    R.pv[v][4] = Q.zu[g];
    R.pv[v][5] = (i < NX) ? ((ie >= 0) ? Q.lam[A.edge_row0[ie] + NW + i] : Q.lam[i]) : 0.0;
    const int iu = is_up ? i - NX : (i >= NA && i < NA + NU ? i - NA : 0);
    const bool hu = i >= NX && i < NA + NU;
    R.pv[v][6] = hu ? (pn >= 0 ? Q.x[A.node_u_off[pn] + iu] : Q.P[A.p_off_uprev + iu] / DOMPC_SU[iu]) : 0.0;
    R.pv[v][7] = 0.0;
    R.pv[v][8] = yi >= 0 ? S_[ES_QV + yi] : 0.0;
    R.pv[v][9] = 0.0;
  }
  if (NE > 0) {
#pragma unroll
    for (int q = 0; q < RN_NLP; ++q) {
      const int it = lane + q * GS;
      R.nl[q] = it < RN_NLN ? node_nl_load(Q, e, it) : 0.0;
    }
  }
#pragma unroll
  for (int q = 0; q < RN_ABP; ++q) {
    const int it = lane + q * GS;
    const int itc = it < RN_ABN ? it : 0;
    const int i = itc / (NA + 1), j = itc % (NA + 1);
    R.ab[q] = (j < NA) ? S_[ES_AB + i * NA + j] : S_[ES_CV + i];
  }
}

// Riccati update of one tree node by one lane group (see riccati_backward).  Leaves P_n, p_n in the group's
// LDS region (RB_PN) and in the node record; `child_staged`: the single child's P_c, p_c are already in
// RB_PC (the group has just computed them while walking up its scenario chain).  R: node_prefetch(n).
DOMPC_DEV inline int riccati_node(const Thr& T, const Prob& Q, int n, double mu, double delta, ldsd* Ld, int lane, int GS,
                                  bool child_staged, const NodePre& R) {
  using namespace rb;
  const KArgs& A = *Q.A;
  double* Nd = Q.ND(n);
  const int cs = A.node_child_start[n], cc = A.node_child_count[n];
  const double rw = node_rweight(Q, n);
  const double rwh = (Q.soc & 2) ? 0.0 : rw;          // weight of the rterm HESSIAN (Prob::soc bit 1)
  long long pc0 = prof_clock();
#if DOMPC_PROFILE
#define DOMPC_PN(i) if (T.prof && T.tid == 0) { const long long pc1 = prof_clock(); T.prof[i] += pc1 - pc0; pc0 = pc1; }
#else
#define DOMPC_PN(i)
#endif
  // ---- pass A: own quadratic (bounds Sigma, rterm, barrier gradients, slack penalty) plus the condensed
  //      blocks of all child edges.  First child + own data come from the prefetched registers, further
  //      children (branching nodes only) are added from global memory.
  double qacc[RN_IPL];
#pragma unroll
  for (int q = 0; q < RN_IPL; ++q) {
    const int it = lane + q * GS;
    const int itc = it < NYT * NYT ? it : 0;
    const int yi = yidx(itc / NYT), yj = yidx(itc % NYT);
    const bool valid = it < NYT * NYT && yi >= 0 && yj >= 0;
    const int idx = valid ? symi(yi, yj, NA) : 0;
    double v = R.qt[q];
    for (int c = 1; c < cc; ++c) v += Q.ES(cs + c)[ES_QT + idx];
    qacc[q] = valid ? v : 0.0;
  }
  // per-variable terms (diagonal + gradient): lanes 0..NYT-1
  double gvv[RN_VPL], dgv[RN_VPL];
#pragma unroll
  for (int v = 0; v < RN_VPL; ++v) {
    const int i0 = lane + v * GS;
    const int i = i0 < NYT ? i0 : 0;
    const int yi = yidx(i);
    const bool is_up = (i >= NX && i < NA);
    const double xv = R.pv[v][0], lo = R.pv[v][1], hi = R.pv[v][2], zlo = R.pv[v][3], zhi = R.pv[v][4];
    const double upv = R.pv[v][6];
    double dg, gv;
    if (is_up) {
      if (RT_CUSTOM) {       // user-defined rterm: d / d u_prev of every edge leaving the node (its Hessian joins the matrix below)
        dg = 0.0; gv = 0.0;
        for (int c = 0; c < cc; ++c) gv += Q.ES(cs + c)[ES_RTUP + (i - NX)];
      } else {
        dg = 2.0 * rwh * DOMPC_RTERM[i - NX];
        gv = -2.0 * rw * DOMPC_RTERM[i - NX] * (xv - upv);                    // xv = u_n of the same input
      }
    } else {
      dg = sigma_of(xv, lo, hi, zlo, zhi) + delta;
      gv = bar_grad(xv, lo, hi, mu, !(Q.soc & 2));
      if (i < NX) gv += (A.node_in_edge[n] >= 0) ? -R.pv[v][5] : R.pv[v][5];
      else if (i < NA + NU) {
        if (!RT_CUSTOM) {    // (user-defined: the (x, u) part of the gradient is in the edges' r_y)
          dg += 2.0 * rwh * DOMPC_RTERM[i - NA];
          gv += 2.0 * rw * DOMPC_RTERM[i - NA] * (xv - upv);
        }
      } else {
        gv += cc * Q.sf * DOMPC_EPS_PEN[i - NA - NU];
      }
    }
    gv += R.pv[v][8];
    if (FREE_ROOT && n == 0 && i0 < NX) gv += Nd[ND_AT + 1 + i];
    if (yi >= 0)
      for (int c = 1; c < cc; ++c) gv += Q.ES(cs + c)[ES_QV + yi];
    gvv[v] = gv;
    dgv[v] = dg;
  }
  if (NE > 0) {
    constexpr int NL_JD = RB_NL, NL_SG = RB_NL + NE * NA, NL_RD = NL_SG + NE, NL_RS = NL_RD + NE, NL_YD = NL_RS + NE;
    for (int c = 0; c < cc; ++c) {
#pragma unroll
      for (int q = 0; q < RN_NLP; ++q) {
        const int it = lane + q * GS;
        if (it < RN_NLN) Ld[RB_NL + it] = (c == 0) ? R.nl[q] : node_nl_load(Q, cs + c, it);
      }
      T.gsync();
#pragma unroll
      for (int v = 0; v < RN_VPL; ++v) {
        const int i0 = lane + v * GS;
        const int i = i0 < NYT ? i0 : 0;
        const int yi = yidx(i);
        double gv = gvv[v];
        for (int q = 0; q < NE; ++q) {
          const double sg = Ld[NL_SG + q] + delta;
          double ji = 0.0;
          if (yi >= 0) ji = Ld[NL_JD + q * NA + yi];
          else if (i >= NA + NU && nl_slack(q) == i - NA - NU) { ji = -Q.sgn[(cs + c) * NE1 + q]; gv += ji * Ld[NL_YD + q]; }      // (column of the slack variable in the scaled row sg (d - eps))
          gv += ji * (sg * Ld[NL_RD + q] + Ld[NL_RS + q]);
        }
        gvv[v] = gv;
      }
#pragma unroll
      for (int q = 0; q < RN_IPL; ++q) {
        const int it = lane + q * GS;
        const int itc = it < NYT * NYT ? it : 0;
        const int i = itc / NYT, j = itc % NYT;
        const int yi = yidx(i), yj = yidx(j);
        double v = qacc[q];
        for (int qq = 0; qq < NE; ++qq) {
          const double sg = Ld[NL_SG + qq] + delta;
          double ji = 0.0, jj = 0.0;
          if (yi >= 0) ji = Ld[NL_JD + qq * NA + yi];
          else if (i >= NA + NU && nl_slack(qq) == i - NA - NU) ji = -Q.sgn[(cs + c) * NE1 + qq];
          if (yj >= 0) jj = Ld[NL_JD + qq * NA + yj];
          else if (j >= NA + NU && nl_slack(qq) == j - NA - NU) jj = -Q.sgn[(cs + c) * NE1 + qq];
          v += sg * ji * jj;
        }
        qacc[q] = v;
      }
      T.gsync();
    }
  }
#pragma unroll
  for (int v = 0; v < RN_VPL; ++v) {
    const int i = lane + v * GS;
    if (i < NYT) {
      Ld[RB_QOV + i] = gvv[v];
      Ld[RB_QFV + i] = 0.0;
      Ld[RB_QF + i * NYT + i] = dgv[v];      // diagonal parked in QF, merged below
    }
  }
  T.gsync();
#pragma unroll
  for (int q = 0; q < RN_IPL; ++q) {
    const int it = lane + q * GS;
    if (it < NYT * NYT) {
      const int i = it / NYT, j = it % NYT;
      double v = qacc[q];
      if (i == j) v += Ld[RB_QF + i * NYT + i];
      if (FREE_ROOT && n == 0 && i < NX && j < NX) v += Nd[ND_AT + 1 + NX + i * NX + j];
      if (RT_CUSTOM) {
        // Hessian of the user-defined rterm over (x, u, u_prev), summed over the edges leaving the node
        auto rz = [](int t) { return t < NX ? t : (t < NA ? NA + (t - NX) : (t < NA + NU ? NX + (t - NA) : -1)); };
        const int ri = rz(i), rj = rz(j);
        if (ri >= 0 && rj >= 0)
          for (int c = 0; c < cc; ++c) v += Q.ES(cs + c)[ES_RTH + symi(ri, rj, NR)];
      } else if (i != j) {
        if (i >= NX && i < NA && j == i + NU) v -= 2.0 * rwh * DOMPC_RTERM[i - NX];
        else if (j >= NX && j < NA && i == j + NU) v -= 2.0 * rwh * DOMPC_RTERM[j - NX];
      }
      Ld[RB_QO + it] = v;
    }
  }
  T.gsync();
  for (int it = lane; it < NYT * NYT; it += GS) Ld[RB_QF + it] = 0.0;
  // stage Atilde (y columns) = [[A|B],[0|I]], ctilde = [c;0] and P_c, p_c of child c
  auto stage_child = [&](int c, bool have_pc) {
    const int e = cs + c;
    const double* S_ = Q.ES(e);
    const double* Nc = Q.ND(A.edge_child[e]);
    if (c == 0) {
#pragma unroll
      for (int q = 0; q < RN_ABP; ++q) {
        const int it = lane + q * GS;
        if (it < RN_ABN) {
          const int i = it / (NA + 1), j = it % (NA + 1);
          if (j < NA) Ld[RB_AT + i * NA + j] = R.ab[q];
          else Ld[RB_CT + i] = R.ab[q];
        }
      }
      for (int it = lane; it < (NA - NX) * (NA + 1); it += GS) {
        const int i = NX + it / (NA + 1), j = it % (NA + 1);
        if (j < NA) Ld[RB_AT + i * NA + j] = (j == i) ? 1.0 : 0.0;
        else Ld[RB_CT + i] = 0.0;
      }
    } else {
      for (int it = lane; it < NA * (NA + 1); it += GS) {
        const int i = it / (NA + 1), j = it % (NA + 1);
        if (j < NA) Ld[RB_AT + i * NA + j] = (i < NX) ? S_[ES_AB + i * NA + j] : ((j == i) ? 1.0 : 0.0);
        else Ld[RB_CT + i] = (i < NX) ? S_[ES_CV + i] : 0.0;
      }
    }
    if (!have_pc) {
      for (int it = lane; it < NA * NA; it += GS) Ld[RB_PC + it] = Nc[ND_P + it];
      for (int it = lane; it < NA; it += GS) Ld[RB_PCV + it] = Nc[ND_PV + it];
    }
  };
  stage_child(0, child_staged && cc == 1);
  T.gsync();
  DOMPC_PN(8)
  // ---- children, pass 1: coupling Atilde' P_c Atilde (and Atilde'(P_c ctilde + p_c)) summed into QF
  for (int c = 0; c < cc; ++c) {
    if (c > 0) { stage_child(c, false); T.gsync(); }
    gmm(lane, GS, NA, NA, NA, (double*)(Ld + RB_PC), NA, 1, (double*)(Ld + RB_AT), NA, 1, 0.0, (double*)(Ld + RB_TP), NA);
    for (int i = lane; i < NA; i += GS) {
      double t = Ld[RB_PCV + i];
#pragma unroll
      for (int a = 0; a < NX; ++a) t += Ld[RB_PC + i * NA + a] * Ld[RB_CT + a];
      Ld[RB_TV + i] = t;
    }
    T.gsync();
    gmm(lane, GS, NA, NA, NA, (double*)(Ld + RB_AT), 1, NA, (double*)(Ld + RB_TP), NA, 1, 0.0, (double*)(Ld + RB_ACL), NA);
    for (int i = lane; i < NA; i += GS) {
      double t = 0.0;
#pragma unroll
      for (int a = 0; a < NA; ++a) t += Ld[RB_AT + a * NA + i] * Ld[RB_TV + a];
      Ld[RB_CCL + i] = t;
    }
    T.gsync();
    for (int it = lane; it < NA * (NA + 1); it += GS) {
      const int yi = it / (NA + 1), yj = it % (NA + 1);
      if (yj < NA) Ld[RB_QF + ycol(yi) * NYT + ycol(yj)] += Ld[RB_ACL + yi * NA + yj];
      else Ld[RB_QFV + ycol(yi)] += Ld[RB_CCL + yi];
    }
    T.gsync();
  }
  DOMPC_PN(9)
  // ---- Cholesky of Qvv (QF + QO) and K = -Qvv^-1 Qvx, kv = -Qvv^-1 qv  (one lane per column)
  int bad = 0;
  for (int j = lane; j < NA + 1; j += GS) {
    double L[NV * NV];
    for (int i = 0; i < NV; ++i)
      for (int jj = 0; jj <= i; ++jj) {
        double t = Ld[RB_QF + (NA + i) * NYT + NA + jj] + Ld[RB_QO + (NA + i) * NYT + NA + jj];
        for (int q = 0; q < jj; ++q) t -= L[i * NV + q] * L[jj * NV + q];
        if (i == jj) {
          if (!(t > 0.0)) { bad = 1; t = 1.0; }
          L[i * NV + i] = sqrt(t);
        } else {
          L[i * NV + jj] = t / L[jj * NV + jj];
        }
      }
    double y[NV];
    for (int i = 0; i < NV; ++i) {
      double t = (j < NA) ? Ld[RB_QF + (NA + i) * NYT + j] + Ld[RB_QO + (NA + i) * NYT + j]
                          : Ld[RB_QFV + NA + i] + Ld[RB_QOV + NA + i];
      for (int q = 0; q < i; ++q) t -= L[i * NV + q] * y[q];
      y[i] = t / L[i * NV + i];
    }
    for (int i = NV - 1; i >= 0; --i) {
      double t = y[i];
      for (int q = i + 1; q < NV; ++q) t -= L[q * NV + i] * y[q];
      y[i] = t / L[i * NV + i];
    }
    for (int i = 0; i < NV; ++i) {
      if (j < NA) { Ld[RB_K + i * NA + j] = -y[i]; Nd[ND_K + i * NA + j] = -y[i]; }
      else { Ld[RB_KV + i] = -y[i]; Nd[ND_KV + i] = -y[i]; }
    }
  }
#ifndef DOMPC_HOST_EMU
  bad = __ballot(bad) != 0ull;          // wave-uniform verdict: the callers branch on it (all lanes stay together)
#endif
  T.gsync();
  DOMPC_PN(10)
  // ---- children, pass 2 (closed-loop form):  PN = Lc' QO Lc + sum Acl' P_c Acl ; pn likewise.
  //      Same pass: own part of PN and the closed-loop map of the staged (last) child.
  auto closed_loop = [&]() {
    for (int it = lane; it < NA * (NA + 1); it += GS) {
      const int i = it / (NA + 1), j = it % (NA + 1);
      double t;
      if (j < NA) {
        // Acl over the augmented state (x, u_prev): column j of [Atilde_x | 0] + Atilde_u K
        t = (j < NX) ? Ld[RB_AT + i * NA + j] : 0.0;
#pragma unroll
        for (int u = 0; u < NU; ++u) t += Ld[RB_AT + i * NA + NX + u] * Ld[RB_K + u * NA + j];
        Ld[RB_ACL + i * NA + j] = t;
      } else {
        t = Ld[RB_CT + i];
#pragma unroll
        for (int u = 0; u < NU; ++u) t += Ld[RB_AT + i * NA + NX + u] * Ld[RB_KV + u];
        Ld[RB_CCL + i] = t;
      }
    }
  };
  for (int it = lane; it < NA * (NA + 1); it += GS) {
    const int i = it / (NA + 1), j = it % (NA + 1);
    if (j < NA) {
      double t = Ld[RB_QO + i * NYT + j];
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        t += Ld[RB_QO + i * NYT + NA + q] * Ld[RB_K + q * NA + j];
        t += Ld[RB_K + q * NA + i] * Ld[RB_QO + (NA + q) * NYT + j];
        double t2 = 0.0;
#pragma unroll
        for (int w = 0; w < NV; ++w) t2 += Ld[RB_QO + (NA + q) * NYT + NA + w] * Ld[RB_K + w * NA + j];
        t += Ld[RB_K + q * NA + i] * t2;
      }
      Ld[RB_PN + i * NA + j] = t;
    } else {
      double t = Ld[RB_QOV + i];
#pragma unroll
      for (int w = 0; w < NV; ++w) t += Ld[RB_QO + i * NYT + NA + w] * Ld[RB_KV + w];
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        double t2 = Ld[RB_QOV + NA + q];
#pragma unroll
        for (int w = 0; w < NV; ++w) t2 += Ld[RB_QO + (NA + q) * NYT + NA + w] * Ld[RB_KV + w];
        t += Ld[RB_K + q * NA + i] * t2;
      }
      Ld[RB_PNV + i] = t;
    }
  }
  for (int c = cc - 1; c >= 0; --c) {
    if (c != cc - 1) { stage_child(c, false); T.gsync(); }      // the last child of pass 1 is still staged
    closed_loop();
    T.gsync();
    gmm(lane, GS, NA, NA, NA, (double*)(Ld + RB_PC), NA, 1, (double*)(Ld + RB_ACL), NA, 1, 0.0, (double*)(Ld + RB_TP), NA);
    for (int i = lane; i < NA; i += GS) {
      double t = Ld[RB_PCV + i];
#pragma unroll
      for (int a = 0; a < NA; ++a) t += Ld[RB_PC + i * NA + a] * Ld[RB_CCL + a];
      Ld[RB_TV + i] = t;
    }
    T.gsync();
    gmm(lane, GS, NA, NA, NA, (double*)(Ld + RB_ACL), 1, NA, (double*)(Ld + RB_TP), NA, 1, 1.0, (double*)(Ld + RB_PN), NA);
    for (int i = lane; i < NA; i += GS) {
      double t = 0.0;
#pragma unroll
      for (int a = 0; a < NA; ++a) t += Ld[RB_ACL + a * NA + i] * Ld[RB_TV + a];
      Ld[RB_PNV + i] += t;
    }
    T.gsync();
  }
  for (int it = lane; it < NA * NA; it += GS) Nd[ND_P + it] = Ld[RB_PN + it];
  for (int it = lane; it < NA; it += GS) Nd[ND_PV + it] = Ld[RB_PNV + it];
  T.gsync();
  DOMPC_PN(11)
#undef DOMPC_PN
  return bad;
}

// Cut parent of a sharded tree (a replicated node whose child sub-trees live on different ranks): the node
// update in three phases around two exchanges (SUM over the ranks of the per-node slots in KArgs::xbuf).
//   phase 1: QO/QOV (own terms: rank 0 only; condensed blocks of the children this rank counts) and the
//            coupling QF/QFV = sum Atilde' P_c Atilde of those children            -> slot in region x_c1
//   phase 2: summed QO..QFV -> K, kv (identical on every rank); closed-loop shares sum Acl' P_c Acl
//            of the counted children                                                -> slot in region x_c2
//   phase 3: P_n = Lc' QO Lc + summed shares -> node record (identical on every rank)
// Plain global loads (no register prefetch): at most a few dozen such nodes per factorisation.
DOMPC_DEV inline int riccati_cut_node(const Thr& T, const Prob& Q, int n, double mu, double delta, ldsd* Ld, int lane,
                                      int GS, int phase) {
  using namespace rb;
  const KArgs& A = *Q.A;
  double* Nd = Q.ND(n);
  const int cs = A.node_child_start[n], cc = A.node_child_count[n];
  const int ci = A.node_cut[n];
  double* X1 = A.xbuf + x_c1(A) + ci * CUT1;
  double* X2 = A.xbuf + x_c2(A) + ci * CUT2;
  auto counted = [&](int c) { return sh_cnt(A, mk_e(A, cs + c)); };
  auto stage_child = [&](int c) {
    const int e = cs + c;
    const double* S_ = Q.ES(e);
    const double* Nc = Q.ND(A.edge_child[e]);
    for (int it = lane; it < NA * (NA + 1); it += GS) {
      const int i = it / (NA + 1), j = it % (NA + 1);
      if (j < NA) Ld[RB_AT + i * NA + j] = (i < NX) ? S_[ES_AB + i * NA + j] : ((j == i) ? 1.0 : 0.0);
      else Ld[RB_CT + i] = (i < NX) ? S_[ES_CV + i] : 0.0;
    }
    for (int it = lane; it < NA * NA; it += GS) Ld[RB_PC + it] = Nc[ND_P + it];
    for (int it = lane; it < NA; it += GS) Ld[RB_PCV + it] = Nc[ND_PV + it];
  };
  if (phase == 1) {
    const bool own = A.shard_rank == 0;
    const double rw = node_rweight(Q, n);
    const double rwh = (Q.soc & 2) ? 0.0 : rw;
    const int xo = A.node_x_off[n], uo = A.node_u_off[n];
    const int eo = NS > 0 ? A.node_eps_off[n] : -1;
    const int ie = A.node_in_edge[n];
    double utmp[NU];
    const double* up = uprev_ptr(Q, n, Q.x, utmp);
    for (int i = lane; i < NYT; i += GS) {
      const int yi = yidx(i);
      const bool is_up = (i >= NX && i < NA);
      const int g = (i < NX) ? xo + i : (is_up ? uo + (i - NX) : (i < NA + NU ? uo + (i - NA) : eo + (i - NA - NU)));
      double dg = 0.0, gv = 0.0;
      if (own) {
        const double xv = Q.x[g], lo = Q.lb[g], hi = Q.ub[g];
        if (is_up) {
          dg = 2.0 * rwh * DOMPC_RTERM[i - NX];
          gv = -2.0 * rw * DOMPC_RTERM[i - NX] * (xv - up[i - NX]);
        } else {
          dg = sigma_of(xv, lo, hi, Q.zl[g], Q.zu[g]) + delta;
          gv = bar_grad(xv, lo, hi, mu, !(Q.soc & 2));
          if (i < NX) gv += (ie >= 0) ? -Q.lam[A.edge_row0[ie] + NW + i] : Q.lam[i];
          else if (i < NA + NU) {
            dg += 2.0 * rwh * DOMPC_RTERM[i - NA];
            gv += 2.0 * rw * DOMPC_RTERM[i - NA] * (xv - up[i - NA]);
          } else {
            gv += cc * Q.sf * DOMPC_EPS_PEN[i - NA - NU];
          }
        }
      }
      for (int c = 0; c < cc; ++c) {
        if (!counted(c)) continue;
        const int e = cs + c;
        const double* S_ = Q.ES(e);
        if (yi >= 0) gv += S_[ES_QV + yi];
        if (NE > 0) {
          const double* yd = Q.lam + A.edge_row0[e] + NW + NX;
          for (int q = 0; q < NE; ++q) {
            const double sg = S_[ES_SIGS + q] + delta;
            double ji = 0.0;
            if (yi >= 0) ji = Q.EW(e, EW_JD + q * NA + yi);
            else if (i >= NA + NU && nl_slack(q) == i - NA - NU) { ji = -Q.sgn[e * NE1 + q]; gv += ji * yd[q]; }
            gv += ji * (sg * S_[ES_RDN + q] + S_[ES_RSN + q]);
          }
        }
      }
      Ld[RB_QOV + i] = gv;
      Ld[RB_QFV + i] = 0.0;
      Ld[RB_QF + i * NYT + i] = dg;      // diagonal parked in QF, merged below
    }
    T.gsync();
    for (int it = lane; it < NYT * NYT; it += GS) {
      const int i = it / NYT, j = it % NYT;
      const int yi = yidx(i), yj = yidx(j);
      double v = 0.0;
      for (int c = 0; c < cc; ++c) {
        if (!counted(c)) continue;
        const int e = cs + c;
        const double* S_ = Q.ES(e);
        if (yi >= 0 && yj >= 0) v += S_[ES_QT + symi(yi, yj, NA)];
        if (NE > 0)
          for (int qq = 0; qq < NE; ++qq) {
            const double sg = S_[ES_SIGS + qq] + delta;
            double ji = 0.0, jj = 0.0;
            if (yi >= 0) ji = Q.EW(e, EW_JD + qq * NA + yi);
            else if (i >= NA + NU && nl_slack(qq) == i - NA - NU) ji = -Q.sgn[e * NE1 + qq];
            if (yj >= 0) jj = Q.EW(e, EW_JD + qq * NA + yj);
            else if (j >= NA + NU && nl_slack(qq) == j - NA - NU) jj = -Q.sgn[e * NE1 + qq];
            v += sg * ji * jj;
          }
      }
      if (i == j) v += Ld[RB_QF + i * NYT + i];
      else if (own && i >= NX && i < NA && j == i + NU) v -= 2.0 * rwh * DOMPC_RTERM[i - NX];
      else if (own && j >= NX && j < NA && i == j + NU) v -= 2.0 * rwh * DOMPC_RTERM[j - NX];
      Ld[RB_QO + it] = v;
    }
    T.gsync();
    for (int it = lane; it < NYT * NYT; it += GS) Ld[RB_QF + it] = 0.0;
    T.gsync();
    for (int c = 0; c < cc; ++c) {
      if (!counted(c)) continue;
      stage_child(c);
      T.gsync();
      gmm(lane, GS, NA, NA, NA, (double*)(Ld + RB_PC), NA, 1, (double*)(Ld + RB_AT), NA, 1, 0.0, (double*)(Ld + RB_TP), NA);
      for (int i = lane; i < NA; i += GS) {
        double t = Ld[RB_PCV + i];
        for (int a = 0; a < NX; ++a) t += Ld[RB_PC + i * NA + a] * Ld[RB_CT + a];
        Ld[RB_TV + i] = t;
      }
      T.gsync();
      gmm(lane, GS, NA, NA, NA, (double*)(Ld + RB_AT), 1, NA, (double*)(Ld + RB_TP), NA, 1, 0.0, (double*)(Ld + RB_ACL), NA);
      for (int i = lane; i < NA; i += GS) {
        double t = 0.0;
        for (int a = 0; a < NA; ++a) t += Ld[RB_AT + a * NA + i] * Ld[RB_TV + a];
        Ld[RB_CCL + i] = t;
      }
      T.gsync();
      for (int it = lane; it < NA * (NA + 1); it += GS) {
        const int yi = it / (NA + 1), yj = it % (NA + 1);
        if (yj < NA) Ld[RB_QF + ycol(yi) * NYT + ycol(yj)] += Ld[RB_ACL + yi * NA + yj];
        else Ld[RB_QFV + ycol(yi)] += Ld[RB_CCL + yi];
      }
      T.gsync();
    }
    for (int it = lane; it < NYT * NYT; it += GS) {
      X1[it] = Ld[RB_QO + it];
      X1[NYT * NYT + NYT + it] = Ld[RB_QF + it];
    }
    for (int i = lane; i < NYT; i += GS) {
      X1[NYT * NYT + i] = Ld[RB_QOV + i];
      X1[2 * NYT * NYT + NYT + i] = Ld[RB_QFV + i];
    }
    T.gsync();
    return 0;
  }
  // phases 2 and 3 start from the summed quadratic
  for (int it = lane; it < NYT * NYT; it += GS) {
    Ld[RB_QO + it] = X1[it];
    Ld[RB_QF + it] = X1[NYT * NYT + NYT + it];
  }
  for (int i = lane; i < NYT; i += GS) {
    Ld[RB_QOV + i] = X1[NYT * NYT + i];
    Ld[RB_QFV + i] = X1[2 * NYT * NYT + NYT + i];
  }
  T.gsync();
  int bad = 0;
  if (phase == 2) {
    // Cholesky of Qvv (QF + QO) and K = -Qvv^-1 Qvx, kv = -Qvv^-1 qv  (one lane per column)
    for (int j = lane; j < NA + 1; j += GS) {
      double L[NV * NV];
      for (int i = 0; i < NV; ++i)
        for (int jj = 0; jj <= i; ++jj) {
          double t = Ld[RB_QF + (NA + i) * NYT + NA + jj] + Ld[RB_QO + (NA + i) * NYT + NA + jj];
          for (int q = 0; q < jj; ++q) t -= L[i * NV + q] * L[jj * NV + q];
          if (i == jj) {
            if (!(t > 0.0)) { bad = 1; t = 1.0; }
            L[i * NV + i] = sqrt(t);
          } else {
            L[i * NV + jj] = t / L[jj * NV + jj];
          }
        }
      double y[NV];
      for (int i = 0; i < NV; ++i) {
        double t = (j < NA) ? Ld[RB_QF + (NA + i) * NYT + j] + Ld[RB_QO + (NA + i) * NYT + j]
                            : Ld[RB_QFV + NA + i] + Ld[RB_QOV + NA + i];
        for (int q = 0; q < i; ++q) t -= L[i * NV + q] * y[q];
        y[i] = t / L[i * NV + i];
      }
      for (int i = NV - 1; i >= 0; --i) {
        double t = y[i];
        for (int q = i + 1; q < NV; ++q) t -= L[q * NV + i] * y[q];
        y[i] = t / L[i * NV + i];
      }
      for (int i = 0; i < NV; ++i) {
        if (j < NA) { Ld[RB_K + i * NA + j] = -y[i]; Nd[ND_K + i * NA + j] = -y[i]; }
        else { Ld[RB_KV + i] = -y[i]; Nd[ND_KV + i] = -y[i]; }
      }
    }
    for (int it = lane; it < NA * NA; it += GS) Ld[RB_PN + it] = 0.0;
    for (int it = lane; it < NA; it += GS) Ld[RB_PNV + it] = 0.0;
    T.gsync();
    for (int c = 0; c < cc; ++c) {
      if (!counted(c)) continue;
      stage_child(c);
      T.gsync();
      for (int it = lane; it < NA * (NA + 1); it += GS) {         // closed-loop map of this child
        const int i = it / (NA + 1), j = it % (NA + 1);
        double t;
        if (j < NA) {
          t = (j < NX) ? Ld[RB_AT + i * NA + j] : 0.0;
          for (int u = 0; u < NU; ++u) t += Ld[RB_AT + i * NA + NX + u] * Ld[RB_K + u * NA + j];
          Ld[RB_ACL + i * NA + j] = t;
        } else {
          t = Ld[RB_CT + i];
          for (int u = 0; u < NU; ++u) t += Ld[RB_AT + i * NA + NX + u] * Ld[RB_KV + u];
          Ld[RB_CCL + i] = t;
        }
      }
      T.gsync();
      gmm(lane, GS, NA, NA, NA, (double*)(Ld + RB_PC), NA, 1, (double*)(Ld + RB_ACL), NA, 1, 0.0, (double*)(Ld + RB_TP), NA);
      for (int i = lane; i < NA; i += GS) {
        double t = Ld[RB_PCV + i];
        for (int a = 0; a < NA; ++a) t += Ld[RB_PC + i * NA + a] * Ld[RB_CCL + a];
        Ld[RB_TV + i] = t;
      }
      T.gsync();
      gmm(lane, GS, NA, NA, NA, (double*)(Ld + RB_ACL), 1, NA, (double*)(Ld + RB_TP), NA, 1, 1.0, (double*)(Ld + RB_PN), NA);
      for (int i = lane; i < NA; i += GS) {
        double t = 0.0;
        for (int a = 0; a < NA; ++a) t += Ld[RB_ACL + a * NA + i] * Ld[RB_TV + a];
        Ld[RB_PNV + i] += t;
      }
      T.gsync();
    }
    for (int it = lane; it < NA * NA; it += GS) X2[it] = Ld[RB_PN + it];
    for (int it = lane; it < NA; it += GS) X2[NA * NA + it] = Ld[RB_PNV + it];
    T.gsync();
    return bad;
  }
  // phase 3: own congruence Lc' QO Lc with the stored gains, plus the summed closed-loop shares
  for (int it = lane; it < NV * NA; it += GS) Ld[RB_K + it] = Nd[ND_K + it];
  for (int it = lane; it < NV; it += GS) Ld[RB_KV + it] = Nd[ND_KV + it];
  T.gsync();
  for (int it = lane; it < NA * (NA + 1); it += GS) {
    const int i = it / (NA + 1), j = it % (NA + 1);
    if (j < NA) {
      double t = Ld[RB_QO + i * NYT + j];
      for (int q = 0; q < NV; ++q) {
        t += Ld[RB_QO + i * NYT + NA + q] * Ld[RB_K + q * NA + j];
        t += Ld[RB_K + q * NA + i] * Ld[RB_QO + (NA + q) * NYT + j];
        double t2 = 0.0;
        for (int w = 0; w < NV; ++w) t2 += Ld[RB_QO + (NA + q) * NYT + NA + w] * Ld[RB_K + w * NA + j];
        t += Ld[RB_K + q * NA + i] * t2;
      }
      Nd[ND_P + i * NA + j] = t + X2[i * NA + j];
    } else {
      double t = Ld[RB_QOV + i];
      for (int w = 0; w < NV; ++w) t += Ld[RB_QO + i * NYT + NA + w] * Ld[RB_KV + w];
      for (int q = 0; q < NV; ++q) {
        double t2 = Ld[RB_QOV + NA + q];
        for (int w = 0; w < NV; ++w) t2 += Ld[RB_QO + (NA + q) * NYT + NA + w] * Ld[RB_KV + w];
        t += Ld[RB_K + q * NA + i] * t2;
      }
      Nd[ND_PV + i] = t + X2[NA * NA + i];
    }
  }
  T.gsync();
  return 0;
}

}  // namespace dompc
#include "dompc_riccati16.h"
namespace dompc {

DOMPC_PHASE int riccati_backward(const Thr& T, const Prob& Q, double mu, double delta) {
#ifndef DOMPC_HOST_EMU
  // register-resident matrix-core recursion (dompc_riccati16.h) unless the model is too large for one tile; the generic
  // LDS-staged path below then is dead code on the device and its working set is not part of the LDS pool
  if constexpr (R16_ENABLED) return r16::backward(T, Q, mu, delta);
#endif
  // One group of lanes (a wavefront) per tree node, the node's matrices staged in the group's LDS region:
  //   RB_QO  own quadratic of the node over (x, u_prev, u, eps)       (NYT x NYT) + gradient
  //   RB_QF  the same plus the children's value functions (coupling)  -> K = -Qvv^-1 Qvx
  //   value function in closed-loop ("Joseph") form  P = Lc' QO Lc + sum_e Acl' P_c Acl,  Lc = [I;K],
  //   Acl = Atilde Lc: the huge Sigma entries of active state bounds inside P_c meet closed-loop maps
  //   that vanish in the constrained directions instead of being cancelled against each other
  //   (Qxx - Qxv Qvv^-1 Qvx floors the KKT residual at ~Sigma_max*eps).
  // Below the robust horizon (stage >= chain_level) every node has one child of the same scenario index:
  // a group walks its scenario chain from the leaf upwards without any barrier and keeps P_c in LDS.
  // The branching part of the tree is processed level by level with a barrier in between.
  using namespace rb;
  const KArgs& A = *Q.A;
  static_assert(!RB_IN_LDS || RB_SIZE <= EL_SIZE, "node working set must fit the per-group LDS region");
  const int GS = T.gs, ng = T.nt / GS, gid = group_index(T.tid, GS), lane = T.tid % GS;
  ldsd* Ld = T.edge_lds + (int64_t)(T.ltid / GS) * EL_SIZE;
  // The failure flag is read by every thread after a barrier and reset here by thread 0.  When the caller repeats the
  // factorisation (inertia correction) a fast wavefront could reset it before a slow one had read the verdict of the
  // previous pass - the wavefronts then disagree about "failed" and the workgroup falls apart (garbage steps or a
  // barrier that never completes; seen as a timing-dependent failure of a 37-problem batch).  Hence the barrier
  // BEFORE the reset: every thread is past its last read of the previous pass.
  const int FSET = T.flag_begin(0);
  {
    // leaves: P = sf*omega*Hm + Sigma_x, p = sf*omega*gm - nu_in + barrier
    const int n0 = A.level_node_start[A.N], n1 = A.level_node_start[A.N + 1];
    for (int it = T.tid; it < (n1 - n0) * NA * (NA + 1); it += T.nt) {
      const int n = n0 + it / (NA * (NA + 1));
      if (!mk_n(A, n)) continue;
      const int r = it % (NA * (NA + 1));
      const int i = r / (NA + 1), j = r % (NA + 1);
      double* Nd = Q.ND(n);
      const int ie = A.node_in_edge[n];
      const double* S_ = Q.ES(ie);
      const int xo = A.node_x_off[n];
      if (j < NA) {
        double v = 0.0;
        if (i < NX && j < NX) {
          v = S_[ES_MH + i * NX + j];
          if (i == j) v += sigma_of(Q.x[xo + i], Q.lb[xo + i], Q.ub[xo + i], Q.zl[xo + i], Q.zu[xo + i]) + delta;
        }
        Nd[ND_P + i * NA + j] = v;
      } else {
        double v = 0.0;
        if (i < NX)
          v = S_[ES_MG + i] - Q.lam[A.edge_row0[ie] + NW + i] + bar_grad(Q.x[xo + i], Q.lb[xo + i], Q.ub[xo + i], mu, !(Q.soc & 2));
        Nd[ND_PV + i] = v;
      }
    }
    T.sync();
  }
  const int cl = A.chain_level < A.N ? A.chain_level : A.N;
  {
    // scenario chains: stages N-1 ... chain_level, node (k, s) -> parent (k-1, s)
    const int S = A.level_node_start[A.N + 1] - A.level_node_start[A.N];
    for (int s_ = gid; s_ < S; s_ += ng) {
      if (!mk_n(A, A.level_node_start[A.N] + s_)) continue;      // another rank's sub-tree
      bool staged = false;
      NodePre R;
      if (A.N - 1 >= cl) node_prefetch(Q, A.level_node_start[A.N - 1] + s_, delta, lane, GS, R);
      for (int k = A.N - 1; k >= cl; --k) {
        NodePre Rn;              // the parent's operands: in flight while this node is updated
        if (k > cl) node_prefetch(Q, A.level_node_start[k - 1] + s_, delta, lane, GS, Rn);
        if (staged) {            // P of the node just finished becomes P_c of its parent
          for (int it = lane; it < NA * NA; it += GS) Ld[RB_PC + it] = Ld[RB_PN + it];
          for (int it = lane; it < NA; it += GS) Ld[RB_PCV + it] = Ld[RB_PNV + it];
          T.gsync();
        }
        if (riccati_node(T, Q, A.level_node_start[k] + s_, mu, delta, Ld, lane, GS, staged, R)) { T.fset(0, FSET); break; }
        staged = true;
        if (k > cl) R = Rn;
      }
    }
    T.sync();
    if ((T.fget(0) == FSET) && !sh_on(A)) return 1;      // (sharded: the flag is only known to this rank until the cut exchange)
  }
  for (int k = cl - 1; k >= 0; --k) {
    const int n0 = A.level_node_start[k], n1 = A.level_node_start[k + 1];
    if (sh_on(A) && k == A.cut_level - 1) {
      // cut parents: their child sub-trees are spread over the ranks -> two exchanges (riccati_cut_node)
      for (int n = n0 + gid; n < n1; n += ng) riccati_cut_node(T, Q, n, mu, delta, Ld, lane, GS, 1);
      T.xchg(x_c1(A), A.n_cut * CUT1);
      for (int n = n0 + gid; n < n1; n += ng)
        if (riccati_cut_node(T, Q, n, mu, delta, Ld, lane, GS, 2)) T.fset(0, FSET);
      T.sync();
      double* fl = A.xbuf + x_c2(A) + A.n_cut * CUT2;          // failure flags of all ranks ride along
      for (int w = T.tid; w < A.shard_world; w += T.nt) fl[w] = (w == A.shard_rank && (T.fget(0) == FSET)) ? 1.0 : 0.0;
      T.xchg(x_c2(A), A.n_cut * CUT2 + A.shard_world);
      for (int n = n0 + gid; n < n1; n += ng) riccati_cut_node(T, Q, n, mu, delta, Ld, lane, GS, 3);
      int bad = 0;
      for (int w = 0; w < A.shard_world; ++w) bad |= (fl[w] != 0.0);
      T.sync();
      if (bad) return 1;
      continue;
    }
    for (int n = n0 + gid; n < n1; n += ng) {
      if (!mk_n(A, n)) continue;
      NodePre R;
      node_prefetch(Q, n, delta, lane, GS, R);
      if (riccati_node(T, Q, n, mu, delta, Ld, lane, GS, false, R)) T.fset(0, FSET);
    }
    T.sync();
    if ((T.fget(0) == FSET) && (!sh_on(A) || k < A.cut_level - 1)) return 1;
  }
  if (FREE_ROOT) {
    // free initial state: its step minimises the root's value function (which holds the arrival cost),
    // P_xx dx = -p_x; P_xx must be positive definite (inertia of the whole system) - else the caller raises delta_w
    T.sync();
    if (T.tid == 0) {
      double* Nd = Q.ND(0);
      constexpr int N1 = NX > 0 ? NX : 1;
      double L[N1 * N1], y[N1];
      bool bad = false;
      for (int i = 0; i < NX; ++i)
        for (int j = 0; j <= i; ++j) {
          double t = Nd[ND_P + i * NA + j];
          for (int q = 0; q < j; ++q) t -= L[i * NX + q] * L[j * NX + q];
          if (i == j) {
            if (!(t > 0.0)) { bad = true; t = 1.0; }
            L[i * NX + i] = sqrt(t);
          } else {
            L[i * NX + j] = t / L[j * NX + j];
          }
        }
      for (int i = 0; i < NX; ++i) {
        double t = -Nd[ND_PV + i];
        for (int q = 0; q < i; ++q) t -= L[i * NX + q] * y[q];
        y[i] = t / L[i * NX + i];
      }
      for (int i = NX - 1; i >= 0; --i) {
        double t = y[i];
        for (int q = i + 1; q < NX; ++q) t -= L[q * NX + i] * y[q];
        y[i] = t / L[i * NX + i];
      }
      for (int a = 0; a < NA; ++a) Nd[ND_DXT + a] = (a < NX) ? y[a] : 0.0;
      if (bad) T.fset(0, FSET);
    }
    T.sync();
    if ((T.fget(0) == FSET)) return 1;
  }
  return 0;
}
