// dompc_dae.h - edge phases of the structured IPM for models WITH algebraic states (DAE, `_z`): dense, general path.
// Included by dompc_kernel.h (inside namespace dompc, after the helpers of the optimised paths).
//
// Reference (what is restated): the interval function of Optimizer._setup_discretization
// (/root/reference/do_mpc/optimizer.py:905-983 - per finite element: algebraic rows at point 0, then per collocation point
// the collocation rows h f(x_ij, u, z_ij) - sum_r C[r,j] x_ir and the algebraic rows a(x_ij, u, z_ij), then the
// continuity rows; discrete models: [alg ; rhs], :820-824), the variable block `_z[k][s][c]` (_mpc.py:1130) and its use in
// the dynamics / stage cost / nl_cons (_mpc.py:1213, 1241, 1252: `_z[k, child, :]`, `_z[k, s, -1]`, `_z[k, s, 0]`).
//
// The algebraic states of an interval are edge unknowns like its collocation states: w = [x slots (M nx) | z slots (MZ nz)],
// eliminated through the edge's own square constraint block G_w (NW x NW, rows in the reference's order).  Everything is
// assembled DENSE in the wavefront's LDS region - the extended matrix [G_w | G_y | r], the edge Hessian over [w | y] -
// by scatter through index functions, inverted in place by Gauss-Jordan with partial pivoting and condensed by plain triple loops:
// this path is about coverage of the reference's DAE models (double inverted pendulum: 36 unknowns per edge), not about the
// benchmark; the node level (tree Riccati recursion) is unchanged, z never reaches it.
#pragma once

// z block of edge e inside opt_x (_mpc.py:1126-1134: `_z` follows `_x`; repeat [N][S][max(M,1)])
DOMPC_DEV inline int edge_zoff(const KArgs& A, int e) {
  const int k = A.edge_level[e];
  const int S = A.level_node_start[A.N + 1] - A.level_node_start[A.N];
  const int c = A.edge_child[e] - A.level_node_start[k + 1];
  return (A.N + 1) * S * (M + 1) * NX + ((k * S + c) * (MZ > 0 ? MZ : 1)) * NZ;
}
// opt_x index of edge unknown c (x slots first, then z slots)
DOMPC_DEV inline int wvar(int woff, int zoff, int c) { return c < NWX ? woff + c : zoff + (c - NWX); }

namespace dae {
constexpr int NWY = NW + NA;                                  // [w | y]
constexpr int DELR = NZ + DEG * (NX + NZ) + NX;               // rows of one finite element
constexpr int ELR1 = DELR > 0 ? DELR : 1;
constexpr int NXZ1 = NX + NZ > 0 ? NX + NZ : 1;
// LDS working set of one edge (doubles)
constexpr int DG_NC = NW + NA + 1;                            // [G_w | G_y | r]  ->  [G_w^-1 | -W | -w0]  (inverted in place)
constexpr int DG_G = 0;
constexpr int DG_LD = DG_NC + 1;                               // row stride: one more column that stays zero
constexpr int DG_H = DG_G + NW * DG_LD;                       // NWY x NWY edge Hessian
constexpr int DG_T = DG_H + NWY * NWY;                        // NWY x (NA + 2):  Hfull [Z | z0 | 0] + [0 | r_w | b]
constexpr int DG_GL = DG_T + NWY * (NA + 2);                  // constraint part of the Lagrangian gradient w.r.t. [w | y]
constexpr int DG_GF = DG_GL + NWY;                            // objective gradient w.r.t. [w | y]
constexpr int DG_RW = DG_GF + NWY;
constexpr int DG_SG = DG_RW + NW;
constexpr int DG_BB = DG_SG + NW;
constexpr int DG_EW = DG_BB + NW;                             // NX x NW: end-point rows w.r.t. w
constexpr int DG_EY = DG_EW + NX * NW;                        // NX x NA: ... w.r.t. y (discrete models)
constexpr int DG_RE = DG_EY + NX * NA;                        // NX: end-point residual
constexpr int DG_JDW = DG_RE + NX;                            // NE x NW
constexpr int DG_JDY = DG_JDW + NE * NW;                      // NE x NA
constexpr int DG_CK = DG_JDY + NE * NA;                       // NW: pivot column copy
constexpr int DG_PV = DG_CK + NW;                             // NW: pivot rows of the elimination (the column interchanges that undo them)
constexpr int DG_RT = DG_PV + NW;                             // user-defined rterm of the edge (RT_LEN)
constexpr int DG_SIZE = DG_RT + RT_LEN;
// forward pass
constexpr int DF_DY = 0, DF_DNU = DF_DY + NA, DF_DW = DF_DNU + NX, DF_RHS = DF_DW + NW, DF_DYD = DF_RHS + NW, DF_SIZE = DF_DYD + NE1;

struct Row { int kind, el, j, comp; };      // kind 0: collocation row (el, j >= 1, state), 1: algebraic row (el, j >= 0, eq.), 2: continuity (el, state)
DOMPC_DEV inline Row decode_row(int r) {
  Row R{1, 0, 0, r};
  if (M == 0) return R;                      // discrete: the block is the algebraic rows of the single point
  R.el = r / ELR1;
  int rr = r % ELR1;
  if (rr < NZ) { R.kind = 1; R.j = 0; R.comp = rr; return R; }
  rr -= NZ;
  if (rr < DEG * (NX + NZ)) {
    R.j = rr / NXZ1 + 1;
    const int q = rr % NXZ1;
    if (q < NX) { R.kind = 0; R.comp = q; } else { R.kind = 1; R.comp = q - NX; }
    return R;
  }
  R.kind = 2; R.j = DEG; R.comp = rr - DEG * (NX + NZ);
  return R;
}
DOMPC_DEV inline int point_of(int el, int j) { return M == 0 ? 0 : el * (DEG + 1) + j; }
// index inside [w | y] of input i = (x, u, z) of the point function at point (el, j)
DOMPC_DEV inline int vtarget(int el, int j, int i) {
  if (i < NX) return (M == 0 || (el == 0 && j == 0)) ? NW + i : slot_of(el, j) * NX + i;
  if (i < NA) return NW + i;
  return NWX + point_of(el, j) * NZ + (i - NA);
}
// inputs of the stage cost: (x_n, u, z of the LAST point), of nl_cons: (x_n, u, z of the FIRST point)  (_mpc.py:1252, 1241)
DOMPC_DEV inline int vtarget_stage(int i, bool last_z) {
  if (LT_END && last_z && i < NX) return (M - 1) * NX + i;        // (estimator: the stage cost reads the END state of the interval)
  if (i < NA) return NW + i;
  return NWX + (last_z ? (MZ - 1) : 0) * NZ + (i - NA);
}
// inputs of nl_cons evaluation `blk`: (x_n, u, z of the first point), or with nl_cons_check_colloc_points (x slot blk, u, z slot blk)
DOMPC_DEV inline int vtarget_nl(int blk, int i) {
  if (!NL_COLLOC) return vtarget_stage(i, false);
  if (i < NX) return nl_pt(blk) * NX + i;
  if (i < NA) return NW + i;
  return NWX + nl_pt(blk) * NZ + (i - NA);
}
}  // namespace dae

// model evaluation of one work item of a DAE model (eval_models): dense records, multipliers gathered per point
DOMPC_DEV inline void dae_eval_item(const Prob& Q, int kind, int e, int j) {
  using namespace dae;
  const KArgs& A = *Q.A;
  const int n = A.edge_parent[e], cn = A.edge_child[e], k = A.edge_level[e];
  const double* xn = Q.x + A.node_x_off[n];
  const double* un = Q.x + A.node_u_off[n];
  const double* w = Q.x + A.edge_w_off[e];
  const double* zb = Q.x + edge_zoff(A, e);
  const double* pp = Q.P + A.p_off_p + A.edge_pidx[e] * NP;
  const double* tvp = Q.P + A.p_off_tvp + k * NTVP;
  const int row0 = A.edge_row0[e];
  double* mo = Q.MO(e);
  if (kind == 0) {
    const int el = (M == 0) ? 0 : j / (DEG + 1), jj = (M == 0) ? 0 : j % (DEG + 1);
    const double* xp = (M == 0 || (el == 0 && jj == 0)) ? xn : w + slot_of(el, jj) * NX;
    double lamv[NF > 0 ? NF : 1];
    if (M == 0) {
      for (int a = 0; a < NX; ++a) lamv[a] = Q.lam[row0 + NW + a];          // rows f - x_c (end-point rows)
      for (int b = 0; b < NZ; ++b) lamv[NX + b] = Q.lam[row0 + b];
    } else {
      const int ar = (jj == 0) ? el * DELR : el * DELR + NZ + (jj - 1) * (NX + NZ) + NX;
      for (int a = 0; a < NX; ++a) lamv[a] = (jj == 0) ? 0.0 : Q.lam[row0 + el * DELR + NZ + (jj - 1) * (NX + NZ) + a];
      for (int b = 0; b < NZ; ++b) lamv[NX + b] = Q.lam[row0 + ar + b];
    }
    double* pt = mo + MO_PT + j * PT_STRIDE;
    dompc_dyn(xp, un, zb + j * NZ, tvp, pp, lamv, pt, pt + NF, pt + NF + NF * NAV);
  } else if (kind == 1) {
    lterm_e(Q, e, LT_END ? w + (M - 1) * NX : xn, un, zb + (MZ - 1) * NZ, tvp, pp, mo + MO_LT, mo + MO_LT + 1, mo + MO_LT + 1 + NAV);
#if DOMPC_XTRA_EW
    mo[MO_LT] += dompc_xtra_ew_f(DOMPC_XTRA_EW_ID[e], w, Q.P);      // (value only: gradient and Hessian over w join the edge block, eval_edge_dae)
#endif
  } else if (kind == 2) {
    if (k == A.N - 1)
      mterm_e(Q, e, Q.x + A.node_x_off[cn], Q.P + A.p_off_tvp + (k + 1) * NTVP, pp, mo + MO_MT, mo + MO_MT + 1, mo + MO_MT + 1 + NX);
  } else if (NE > 0) {
    for (int blk = 0; blk < NLB; ++blk) {
      double* o = mo + MO_NL + blk * NL_STRIDE;
      double yds[NEB1];     // (scaled rows sg d(x): the Hessian sum_i lambda_i sg_i hess d_i)
      for (int i = 0; i < NEB; ++i) yds[i] = Q.lam[row0 + NW + NX + blk * NEB + i] * Q.sgn[e * NE1 + blk * NEB + i];
      nlcons_e(Q, e, NL_COLLOC ? w + nl_pt(blk) * NX : xn, un, zb + (NL_COLLOC ? nl_pt(blk) * NZ : 0), tvp, pp, yds,
                   o, o + NEB, o + NEB + NEB * NAV);
    }
  }
}

// constraint residuals + objective share of edge e at the trial point (DAE twin of eval_edge_f)
DOMPC_DEV inline double dae_edge_f(const Prob& Q, int e, const double* xv, const double* sv, double* cv) {
  using namespace dae;
  const KArgs& A = *Q.A;
  const int n = A.edge_parent[e], cn = A.edge_child[e], k = A.edge_level[e];
  const double* xn = xv + A.node_x_off[n];
  const double* un = xv + A.node_u_off[n];
  const double* xc = xv + A.node_x_off[cn];
  const double* w = xv + A.edge_w_off[e];
  const double* zb = xv + edge_zoff(A, e);
  const double* pp = Q.P + A.p_off_p + A.edge_pidx[e] * NP;
  const double* tvp = Q.P + A.p_off_tvp + k * NTVP;
  const int row0 = A.edge_row0[e];
  const double om = A.edge_omega[e] * Q.sf;
  double F[NF > 0 ? NF : 1];
  if (M == 0) {
    dompc_dyn_f(xn, un, zb, tvp, pp, F);
    for (int b = 0; b < NZ; ++b) cv[row0 + b] = F[NX + b];
    for (int a = 0; a < NX; ++a) cv[row0 + NW + a] = F[a] - xc[a];
  } else {
    for (int el = 0; el < NI; ++el) {
      const double* x0 = (el == 0) ? xn : w + slot_of(el, 0) * NX;
      const int rb = row0 + el * DELR;
      for (int jj = 0; jj <= DEG; ++jj) {
        const double* xp = (jj == 0) ? x0 : w + slot_of(el, jj) * NX;
        dompc_dyn_f(xp, un, zb + point_of(el, jj) * NZ, tvp, pp, F);
        if (jj == 0) {
          for (int b = 0; b < NZ; ++b) cv[rb + b] = F[NX + b];
        } else {
          const int r1 = rb + NZ + (jj - 1) * (NX + NZ);
          for (int a = 0; a < NX; ++a) {
            double xp_ = DOMPC_C[0 * (DEG + 1) + jj] * x0[a];
            for (int r = 1; r <= DEG; ++r) xp_ += DOMPC_C[r * (DEG + 1) + jj] * w[slot_of(el, r) * NX + a];
            cv[r1 + a] = F[a] - xp_;
          }
          for (int b = 0; b < NZ; ++b) cv[r1 + NX + b] = F[NX + b];
        }
      }
      const double* xnext = w + next_slot(el) * NX;
      for (int a = 0; a < NX; ++a) {
        double xf = DOMPC_D[0] * x0[a];
        for (int r = 1; r <= DEG; ++r) xf += DOMPC_D[r] * w[slot_of(el, r) * NX + a];
        cv[rb + NZ + DEG * (NX + NZ) + a] = xnext[a] - xf;
      }
    }
    for (int a = 0; a < NX; ++a) cv[row0 + NW + a] = w[(M - 1) * NX + a] - xc[a];
  }
  double obj = om * lterm_f_e(Q, e, LT_END ? w + (M - 1) * NX : xn, un, zb + (MZ - 1) * NZ, tvp, pp);
#if DOMPC_XTRA_EW
  obj += om * dompc_xtra_ew_f(DOMPC_XTRA_EW_ID[e], w, Q.P);
#endif
  if (k == A.N - 1) obj += om * mterm_f_e(Q, e, xc, Q.P + A.p_off_tvp + (k + 1) * NTVP, pp);
  if (RT_CUSTOM) obj += edge_rterm_f(Q, e, xv);
  if (NE > 0) {
    double d[NE1];
    for (int blk = 0; blk < NLB; ++blk)
      nlcons_f_e(Q, e, NL_COLLOC ? w + nl_pt(blk) * NX : xn, un, zb + (NL_COLLOC ? nl_pt(blk) * NZ : 0), tvp, pp, d + blk * NEB);
    const double* eps = (NSE > 0) ? xv + A.node_eps_off[n] : nullptr;
    for (int i = 0; i < NE; ++i) {
      if (nl_slack(i) >= 0) d[i] -= eps[nl_slack(i)];
      d[i] *= Q.sgn[e * NE1 + i];
      cv[row0 + NW + NX + i] = d[i] - sv[e * NE1 + i];
    }
    for (int q = 0; q < NSE; ++q) obj += Q.sf * DOMPC_EPS_PEN[q] * eps[q];
  }
  return obj;
}

#ifndef DOMPC_HOST_EMU
// maximum over the 64 lanes of a wavefront of NON-NEGATIVE values (identity 0), in every lane: row shifts, row broadcasts (DPP), lane 63
__device__ inline double wave_max_nonneg(double v) {
#define DOMPC_DPP_MAX(ctrl, rmask) {                                                                   \
    const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rmask, 0xf, false);        \
    const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rmask, 0xf, false);        \
    v = fmax(v, __hiloint2double(hi_, lo_)); }
  DOMPC_DPP_MAX(0x111, 0xf)      // row_shr:1
  DOMPC_DPP_MAX(0x112, 0xf)      // row_shr:2
  DOMPC_DPP_MAX(0x114, 0xf)      // row_shr:4
  DOMPC_DPP_MAX(0x118, 0xf)      // row_shr:8   -> lane 15 of every row: maximum of the row
  DOMPC_DPP_MAX(0x142, 0xa)      // row_bcast:15 into rows 1 and 3
  DOMPC_DPP_MAX(0x143, 0xc)      // row_bcast:31 into rows 2 and 3 -> lane 63: maximum of the wavefront
#undef DOMPC_DPP_MAX
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
#endif

// ================================================================================================
// Derivative evaluation + condensing of one edge of a DAE model (dense twin of eval_edge_coop; same outputs: the shared
// record ES, the forward record EW, c / gf / rd of the edge's rows and unknowns).  Returns 1 if G_w is singular.
DOMPC_PHASE int eval_edge_dae(const Thr& T, const Prob& Q, int e, double mu, int lane, int GS, ldsd* Ld) {
  using namespace dae;
  const KArgs& A = *Q.A;
  if (e < 0) return 0;
  const int n = A.edge_parent[e], cn = A.edge_child[e], k = A.edge_level[e];
  const double* xn = Q.x + A.node_x_off[n];
  const double* xc = Q.x + A.node_x_off[cn];
  const int woff = A.edge_w_off[e], zoff = edge_zoff(A, e);
  const double* w = Q.x + woff;
  const int row0 = A.edge_row0[e];
  const double om = A.edge_omega[e] * Q.sf;
  const double omh = (Q.soc & 2) ? 0.0 : om;
  const double* lam_e = Q.lam + row0;
  const double* nu_e = Q.lam + row0 + NW;
  const double* yd = Q.lam + row0 + NW + NX;
  double* S_ = Q.ES(e);
  const double* mo = Q.MO(e);
  const bool last_stage = (k == A.N - 1);
  int fail = 0;
  auto Gm = [&](int r, int c) -> ldsd& { return Ld[DG_G + r * DG_LD + c]; };
  auto Hm = [&](int a, int b) -> ldsd& { return Ld[DG_H + a * NWY + b]; };
#if DOMPC_PROFILE && !defined(DOMPC_HOST_EMU)
  // sub-phase cycle counters of the dense path (slots of the edge counters of the optimised path, tools/gpu_profile.py):
  // 0 clear + rows, 4 Hessian scatter, 5 gradient + per-variable terms, 6 elimination, 1 dynamics + T, 2 condensing, 7 record, 3 tail
  long long pc0_ = clock64();
#define DAE_PH(i) if (threadIdx.x == 0) { const long long pc1_ = clock64(); lds_prof[i] += pc1_ - pc0_; pc0_ = pc1_; }
#else
#define DAE_PH(i)
#endif
  // ---- A: clear
  for (int i = lane; i < DG_SIZE; i += GS) Ld[i] = 0.0;
  T.gsync();
  // ---- B: rows of the block [G_w | G_y | r]: residual, collocation / continuity coefficients, point Jacobians
  for (int r = lane; r < NW; r += GS) {
    const Row R = decode_row(r);
    const double* pt = mo + MO_PT + point_of(R.el, R.j) * PT_STRIDE;
    double res;
    if (R.kind == 2) {
      const double* x0 = (R.el == 0) ? xn : w + slot_of(R.el, 0) * NX;
      double xf = DOMPC_D[0] * x0[R.comp];
      Gm(r, vtarget(R.el, 0, R.comp) < NW ? vtarget(R.el, 0, R.comp) : NW + R.comp) -= DOMPC_D[0];
      for (int q = 1; q <= DEG; ++q) {
        xf += DOMPC_D[q] * w[slot_of(R.el, q) * NX + R.comp];
        Gm(r, slot_of(R.el, q) * NX + R.comp) -= DOMPC_D[q];
      }
      const int ns_ = next_slot(R.el);
      res = w[ns_ * NX + R.comp] - xf;
      Gm(r, ns_ * NX + R.comp) += 1.0;
    } else {
      const int o = (R.kind == 0) ? R.comp : NX + R.comp;         // output of the point function
      res = pt[o];
      for (int i = 0; i < NAV; ++i) {
        const int v = vtarget(R.el, R.j, i);
        Gm(r, v) += pt[NF + o * NAV + i];                          // (v < NW: G_w, v >= NW: G_y column v - NW)
      }
      if (R.kind == 0) {
        const double* x0 = (R.el == 0) ? xn : w + slot_of(R.el, 0) * NX;
        double xp = DOMPC_C[0 * (DEG + 1) + R.j] * x0[R.comp];
        Gm(r, vtarget(R.el, 0, R.comp)) -= DOMPC_C[0 * (DEG + 1) + R.j];
        for (int q = 1; q <= DEG; ++q) {
          xp += DOMPC_C[q * (DEG + 1) + R.j] * w[slot_of(R.el, q) * NX + R.comp];
          Gm(r, slot_of(R.el, q) * NX + R.comp) -= DOMPC_C[q * (DEG + 1) + R.j];
        }
        res -= xp;
      }
    }
    if (Q.soc & 1) res = Q.c[row0 + r];
    else Q.c[row0 + r] = res;
    Gm(r, NW + NA) = res;
  }
  // end-point rows: continuous  w_end - x_c ;  discrete  f(x_n, u, z) - x_c
  for (int a = lane; a < NX; a += GS) {
    double res;
    if (M == 0) {
      const double* pt = mo + MO_PT;
      res = pt[a] - xc[a];
      for (int i = 0; i < NAV; ++i) {
        const int v = vtarget(0, 0, i);
        if (v < NW) Ld[DG_EW + a * NW + v] = pt[NF + a * NAV + i];
        else Ld[DG_EY + a * NA + (v - NW)] = pt[NF + a * NAV + i];
      }
    } else {
      res = w[(M - 1) * NX + a] - xc[a];
      Ld[DG_EW + a * NW + (M - 1) * NX + a] = 1.0;
    }
    if (Q.soc & 1) res = Q.c[row0 + NW + a];
    else Q.c[row0 + NW + a] = res;
    Ld[DG_RE + a] = res;
  }
  // nl_cons Jacobian split into the w part and the y part
  for (int it = lane; it < NE * NAV; it += GS) {
    const int q = it / NAV, i = it % NAV;
    const int v = vtarget_nl(q / NEB1, i);
    const double jv = mo[MO_NL + (q / NEB1) * NL_STRIDE + NEB + (q % NEB1) * NAV + i] * Q.sgn[e * NE1 + q];
    if (v < NW) Ld[DG_JDW + q * NW + v] = jv;
    else Ld[DG_JDY + q * NA + (v - NW)] = jv;
  }
  // objective gradient: stage cost over (x_n, u, z_last); user-defined rterm over (x_n, u) (its u_prev part and its Hessian go
  // to the node level through the record)
  for (int i = lane; i < NAV; i += GS) Ld[DG_GF + vtarget_stage(i, true)] = om * mo[MO_LT + 1 + i];
  if (RT_CUSTOM && lane == 0) edge_rterm_eval(Q, e, Ld + DG_RT);
  T.gsync();
  if (RT_CUSTOM) {
    for (int a = lane; a < NA; a += GS) Ld[DG_GF + NW + a] += Ld[DG_RT + 1 + a];
    edge_rterm_store(Ld + DG_RT, S_, lane, GS);
    T.gsync();
  }
  DAE_PH(0)
  // ---- Hessian of the edge's Lagrangian terms over [w | y]: one function after the other (their supports overlap)
  for (int p = 0; p < NPT_E; ++p) {
    const int el = (M == 0) ? 0 : p / (DEG + 1), jj = (M == 0) ? 0 : p % (DEG + 1);
    const double* Hp = mo + MO_PT + p * PT_STRIDE + NF + NF * NAV;
    for (int it = lane; it < NAV * NAV; it += GS) {
      const int i1 = it / NAV, i2 = it % NAV;
      Hm(vtarget(el, jj, i1), vtarget(el, jj, i2)) += Hp[symi(i1, i2, NAV)];
    }
    T.gsync();
  }
  for (int it = lane; it < NAV * NAV; it += GS) {
    const int i1 = it / NAV, i2 = it % NAV;
    Hm(vtarget_stage(i1, true), vtarget_stage(i2, true)) += omh * mo[MO_LT + 1 + NAV + symi(i1, i2, NAV)];
  }
  T.gsync();
#if DOMPC_XTRA_EW
  // cost terms the user added in the collocation states of this interval (nlp_route.py, kind "ew"): gradient and Hessian entries over w
  if (lane == 0)
    dompc_xtra_ew(DOMPC_XTRA_EW_ID[e], w, Q.P,
                  [&](int i, double v) { Ld[DG_GF + i] += om * v; },
                  [&](int i, int j, double v) { Hm(i, j) += omh * v; if (i != j) Hm(j, i) += omh * v; });
  T.gsync();
#endif
  if (NE > 0) {
    for (int blk = 0; blk < NLB; ++blk) {
      for (int it = lane; it < NAV * NAV; it += GS) {
        const int i1 = it / NAV, i2 = it % NAV;
        Hm(vtarget_nl(blk, i1), vtarget_nl(blk, i2)) += mo[MO_NL + blk * NL_STRIDE + NEB + NEB * NAV + symi(i1, i2, NAV)];
      }
      T.gsync();
    }
  }
  DAE_PH(4)
  // ---- constraint part of the Lagrangian gradient: [G_w G_y]' lambda + [E_w E_y]' nu + [Jd_w Jd_y]' y_d
  for (int v = lane; v < NWY; v += GS) {
    double t = 0.0;
    for (int r = 0; r < NW; ++r) t += Gm(r, v) * lam_e[r];
    for (int a = 0; a < NX; ++a) t += (v < NW ? Ld[DG_EW + a * NW + v] : Ld[DG_EY + a * NA + (v - NW)]) * nu_e[a];
    for (int q = 0; q < NE; ++q) t += (v < NW ? Ld[DG_JDW + q * NW + v] : Ld[DG_JDY + q * NA + (v - NW)]) * yd[q];
    Ld[DG_GL + v] = t;
  }
  T.gsync();
  // per-variable terms of the eliminated unknowns
  for (int c = lane; c < NW; c += GS) {
    const int gi = wvar(woff, zoff, c);
    const double xv = Q.x[gi], l = Q.lb[gi], u = Q.ub[gi], zl_ = Q.zl[gi], zu_ = Q.zu[gi];
    const double t = Ld[DG_GL + c] + Ld[DG_GF + c];
    Q.gf[gi] = Ld[DG_GF + c];
    Q.rd[gi] = t - zl_ + zu_;
    Ld[DG_RW + c] = t + bar_grad(xv, l, u, mu, !(Q.soc & 2));
    Ld[DG_BB + c] = bar_grad(xv, l, u, 1.0);
    Ld[DG_SG + c] = sigma_of(xv, l, u, zl_, zu_) + Q.dsw;
  }
  T.gsync();
  DAE_PH(5)
  // ---- Gauss-Jordan with partial pivoting, IN PLACE: the eliminated column kk takes the column of the inverse that an appended
  // identity would hold (row kk scaled by 1 / pivot, entry (kk, kk) = 1 / pivot, the other rows -a_rk / pivot), so every step
  // touches NW + NA + 1 columns instead of 2 NW + NA + 1; the row interchanges are undone at the end by interchanging the
  // COLUMNS of the inverse in reverse order ((P A)^-1 = A^-1 P')
  // Per step ONE batch of LDS reads - the pivot column (every lane, broadcast reads) and the lane's own column -, the pivot search, the
  // row interchange and the rank-1 update in registers, one batch of writes: the first version walked the rows in loops whose every
  // trip waited for its LDS round trip (pivot search, column copy, interchange pass, update: ~10 k cycles per step at 36 unknowns).
#ifndef DOMPC_GJ_REGS_MAX
#define DOMPC_GJ_REGS_MAX 48        // largest block whose pivot steps run in registers (tests: 0 = the row loops for every block)
#endif
  constexpr bool GJ_REGS = NW <= DOMPC_GJ_REGS_MAX;      // (2 NW doubles in registers; larger blocks keep the row loops)
  constexpr int NW1_ = NW > 0 ? NW : 1;
  for (int kk = 0; kk < NW; ++kk) {
    if constexpr (GJ_REGS) {
      // reads of the step in one batch: the pivot column (every lane, broadcast reads), this lane's own column (the owner of the pivot
      // column reads the zero column behind the block instead: its column starts from the unit column of the appended identity)
      double ck[NW1_], g[NW1_];
#pragma unroll
      for (int r = 0; r < NW; ++r) ck[r] = Gm(r, kk);
      int c = lane;
      bool own = (c == kk);
      if (c < DG_NC) {
#pragma unroll
        for (int r = 0; r < NW; ++r) g[r] = Gm(r, own ? DG_NC : c);
      }
      double g_k = (c < DG_NC) ? (double)Gm(kk, own ? DG_NC : c) : 0.0;
      // pivot row: the first row >= kk that attains the largest magnitude
      int pr = NW;
      double best;
#ifndef DOMPC_HOST_EMU
      {
        // lane r holds |a_rk| (0 outside kk <= r < NW), maximum over the wavefront by DPP shifts, first lane that attains it by ballot:
        // ~30 instructions instead of a chain over the NW rows in every lane
        const bool valid = lane >= kk && lane < NW;
        const double a = valid ? fabs((double)Gm(valid ? lane : kk, kk)) : 0.0;
        best = wave_max_nonneg(a);
        const unsigned long long m = __ballot(valid && a == best);
        if (m) pr = (int)__builtin_ctzll(m);
      }
#else
      best = -1.0;
      for (int r = kk; r < NW; ++r) best = fmax(best, fabs(ck[r]));
      for (int r = NW - 1; r >= kk; --r) pr = (fabs(ck[r]) == best) ? r : pr;
#endif
      if (!(best > 1e-300) || pr >= NW) { fail = 1; pr = kk; }      // (NaN in the column: no row attains the "maximum")
      DAE_PH(24)
      const double ck_k = Gm(kk, kk), piv = Gm(pr, kk);            // (entries kk and pr of the pivot column: dynamic rows)
      double g_p = (c < DG_NC) ? (double)Gm(pr, own ? DG_NC : c) : 0.0;
      const double pinv = (fabs(piv) > 1e-300) ? 1.0 / piv : 1.0;
      T.gsync();                                                   // (the owner of column kk overwrites it below)
      if (lane == 0) Ld[DG_PV + kk] = (double)pr;
      DAE_PH(25)
      while (c < DG_NC) {
        const double prow = (own ? 1.0 : g_p) * pinv;               // (row kk after the interchange = the old row pr)
        // every row by the plain formula, then the two rows of the interchange once more (LDS writes of a wavefront land in order):
        // no select per entry
#pragma unroll
        for (int r = 0; r < NW; ++r) Gm(r, c) = fma(-ck[r], prow, g[r]);
        if (pr != kk) Gm(pr, c) = fma(-ck_k, prow, g_k);             // (row pr after the interchange = the old row kk)
        Gm(kk, c) = prow;
        c += GS;
        if (c < DG_NC) {                                             // (blocks wider than the wavefront, host emulation: the next column)
          own = (c == kk);
#pragma unroll
          for (int r = 0; r < NW; ++r) g[r] = Gm(r, own ? DG_NC : c);
          g_k = Gm(kk, own ? DG_NC : c);
          g_p = Gm(pr, own ? DG_NC : c);
        }
      }
      T.gsync();
      DAE_PH(26)
    } else {
      int pr = kk;
      double best = fabs((double)Gm(kk, kk));
      for (int r = kk + 1; r < NW; ++r) {
        const double a = fabs((double)Gm(r, kk));
        if (a > best) { best = a; pr = r; }
      }
      if (!(best > 1e-300)) fail = 1;
      T.gsync();
      if (lane == 0) Ld[DG_PV + kk] = (double)pr;
      if (pr != kk)
        for (int c = lane; c < DG_NC; c += GS) { const double t = Gm(kk, c); Gm(kk, c) = Gm(pr, c); Gm(pr, c) = t; }
      T.gsync();
      const double piv = Gm(kk, kk);
      const double pinv = (fabs(piv) > 1e-300) ? 1.0 / piv : 1.0;
      for (int r = lane; r < NW; r += GS) Ld[DG_CK + r] = Gm(r, kk);
      T.gsync();
      for (int c = lane; c < DG_NC; c += GS) {
        const bool own = (c == kk);
        const double prow = (own ? 1.0 : (double)Gm(kk, c)) * pinv;
        for (int r = 0; r < NW; ++r)
          if (r != kk) Gm(r, c) = fma(-(double)Ld[DG_CK + r], prow, own ? 0.0 : (double)Gm(r, c));
        Gm(kk, c) = prow;
      }
      T.gsync();
    }
  }
  for (int r = lane; r < NW; r += GS)                    // (row r of the inverse: its own sequence of interchanges, no barrier in between)
    for (int kk = NW - 1; kk >= 0; --kk) {
      const int pr = (int)(double)Ld[DG_PV + kk];
      if (pr != kk) { const double t = Gm(r, kk); Gm(r, kk) = Gm(r, pr); Gm(r, pr) = t; }
    }
  T.gsync();
  DAE_PH(27)
  // now: columns 0 .. NW-1 = G_w^-1, columns NW .. NW+NA-1 = G_w^-1 G_y = -W, column NW+NA = G_w^-1 r = -w0
  auto Wm = [&](int r, int c) -> double { return -(double)Gm(r, NW + c); };        // c == NA: w0
  auto Gi = [&](int r, int c) -> double { return (double)Gm(r, c); };
  // ---- linearised dynamics of the interval: [A B] = E_y + E_w W, c~ = r_end + E_w w0; effective nl_cons rows
  for (int it = lane; it < NX * (NA + 1); it += GS) {
    const int a = it / (NA + 1), b = it % (NA + 1);
    double t = (b < NA) ? (double)Ld[DG_EY + a * NA + b] : (double)Ld[DG_RE + a];
    for (int c = 0; c < NW; ++c) t += Ld[DG_EW + a * NW + c] * Wm(c, b);
    if (b < NA) S_[ES_AB + a * NA + b] = t;
    else S_[ES_CV + a] = t;
  }
  for (int it = lane; it < NE * NA; it += GS) {
    const int q = it / NA, b = it % NA;
    double t = Ld[DG_JDY + q * NA + b];
    for (int c = 0; c < NW; ++c) t += Ld[DG_JDW + q * NW + c] * Wm(c, b);
    Q.EW(e, EW_JD + it) = t;
  }
  // ---- T = Hfull [Z | z0 | 0] + [0 | r_w | b],  Z = [W; I], z0 = [w0; 0], Hfull = H + diag(Sigma_w + dsw)
  for (int it = lane; it < NWY * (NA + 2); it += GS) {
    const int v = it / (NA + 2), c = it % (NA + 2);
    double t = 0.0;
    if (c <= NA) {
      for (int q = 0; q < NW; ++q) t += Hm(v, q) * Wm(q, c);
      if (v < NW) t += Ld[DG_SG + v] * Wm(v, c);
      if (c < NA) t += Hm(v, NW + c);
      else if (v < NW) t += Ld[DG_RW + v];
    } else if (v < NW) {
      t = Ld[DG_BB + v];
    }
    Ld[DG_T + v * (NA + 2) + c] = t;
  }
  T.gsync();
  DAE_PH(1)
  // ---- Q~ = Z' T, q~ and W'b
  for (int it = lane; it < NA * (NA + 2); it += GS) {
    const int a = it / (NA + 2), c = it % (NA + 2);
    double t = Ld[DG_T + (NW + a) * (NA + 2) + c];
    for (int q = 0; q < NW; ++q) t += Wm(q, a) * Ld[DG_T + q * (NA + 2) + c];
    if (c < NA) { if (a <= c) S_[ES_QT + symi(a, c, NA)] = t; }
    else if (c == NA) S_[ES_QV + a] = t + (double)Ld[DG_GL + NW + a] + (double)Ld[DG_GF + NW + a];      // q~ + r_y
    else S_[ES_QVB + a] = t;
  }
  for (int a = lane; a < NA; a += GS) {
    S_[ES_RY + a] = (double)Ld[DG_GL + NW + a] + (double)Ld[DG_GF + NW + a];
    S_[ES_GFY + a] = Ld[DG_GF + NW + a];
  }
  DAE_PH(2)
  // ---- forward record
  for (int it = lane; it < NW * NW; it += GS) Q.EW(e, EW_LU + it) = Gi(it / NW, it % NW);
  for (int it = lane; it < NW * NA; it += GS) Q.EW(e, EW_W + it) = Wm(it / NA, it % NA);
  for (int r = lane; r < NW; r += GS) {
    Q.EW(e, EW_W0 + r) = Wm(r, NA);
    Q.EW(e, EW_SIGW + r) = Ld[DG_SG + r];
    Q.EW(e, EW_RW + r) = Ld[DG_RW + r];
  }
  for (int it = lane; it < NW * NWY; it += GS) {
    const int r = it / NWY, v = it % NWY;
    Q.EW(e, EW_HW + it) = Hm(r, v) + (v == r ? (double)Ld[DG_SG + r] : 0.0);
  }
  for (int it = lane; it < NX * NW; it += GS) Q.EW(e, EW_EWJ + it) = Ld[DG_EW + it];
  for (int it = lane; it < NE * NW; it += GS) Q.EW(e, EW_JDW + it) = Ld[DG_JDW + it];
  DAE_PH(7)
  // ---- terminal cost, objective share, nl_cons rows (as in eval_edge_coop, phase 7)
  if (last_stage) {
    for (int a = lane; a < NX; a += GS) S_[ES_MG + a] = om * mo[MO_MT + 1 + a];
    for (int a = lane; a < NX * NX; a += GS) S_[ES_MH + a] = omh * mo[MO_MT + 1 + NX + symi(a / NX, a % NX, NX)];
  }
  if (lane == 0) {
    double obj = om * mo[MO_LT];
    if (last_stage) obj += om * mo[MO_MT];
    if (RT_CUSTOM) obj += Ld[DG_RT];
    if (NE > 0) {
      const double* eps = (NSE > 0) ? Q.x + A.node_eps_off[n] : nullptr;
      for (int i = 0; i < NE; ++i) {
        double d = mo[MO_NL + (i / NEB1) * NL_STRIDE + i % NEB1];
        if (nl_slack(i) >= 0) d -= eps[nl_slack(i)];
        const int si = e * NE1 + i;
        d *= Q.sgn[si];
        const double sv = Q.s[si], l = Q.sl[si], u = Q.su[si];
        double rdn = (Q.soc & 1) ? Q.c[row0 + NW + NX + i] : d - sv;
        if (!(Q.soc & 1)) Q.c[row0 + NW + NX + i] = rdn;
        for (int c = 0; c < NW; ++c) rdn += Ld[DG_JDW + i * NW + c] * Wm(c, NA);       // the row in the reduced variables: + Jd_w w0
        S_[ES_RDN + i] = rdn;
        S_[ES_SIGS + i] = sigma_of(sv, l, u, Q.zsl[si], Q.zsu[si]);
        S_[ES_RSN + i] = -yd[i] + bar_grad(sv, l, u, mu);
      }
      for (int q = 0; q < NSE; ++q) obj += Q.sf * DOMPC_EPS_PEN[q] * eps[q];
    }
    S_[ES_OBJ] = obj;
  }
  T.gsync();
  DAE_PH(3)
#undef DAE_PH
  return fail;
}

// ================================================================================================
// Forward pass of one edge of a DAE model: collocation / algebraic steps dw = W dy + w0 and the multiplier steps of the
// edge's rows  d lambda = G_w^-T [ -(r_w + Hfull_w [dw; dy] + E_w' d nu + Jd_w' d y_d) ].
// In: Ld[DF_DY] (dy of the parent node), Ld[DF_DNU] (d nu of the end-point rows).
DOMPC_DEV inline void forward_edge_dae(const Thr& T, const Prob& Q, int e, double delta, int lane, int GS, ldsd* Ld) {
  using namespace dae;
  const KArgs& A = *Q.A;
  const int n = A.edge_parent[e];
  const int row0 = A.edge_row0[e], woff = A.edge_w_off[e], zoff = edge_zoff(A, e);
  const double* S_ = Q.ES(e);
  for (int r = lane; r < NW; r += GS) {
    double t = Q.EW(e, EW_W0 + r);
    for (int b = 0; b < NA; ++b) t += Q.EW(e, EW_W + r * NA + b) * Ld[DF_DY + b];
    Ld[DF_DW + r] = t;
    Q.dx[wvar(woff, zoff, r)] = t;
  }
  for (int i = lane; i < NE; i += GS) {
    double t = S_[ES_RDN + i];
    for (int b = 0; b < NA; ++b) t += Q.EW(e, EW_JD + i * NA + b) * Ld[DF_DY + b];
    if (!EPS_GLOBAL && nl_slack(i) >= 0) t -= Q.sgn[e * NE1 + i] * Q.dx[A.node_eps_off[n] + nl_slack(i)];
    Q.ds[e * NE1 + i] = t;
    const double dyd = (S_[ES_SIGS + i] + delta) * t + S_[ES_RSN + i];
    Q.dlam[row0 + NW + NX + i] = dyd;
    Ld[DF_DYD + i] = dyd;
  }
  T.gsync();
  for (int r = lane; r < NW; r += GS) {
    double t = Q.EW(e, EW_RW + r);
    for (int v = 0; v < NW; ++v) t += Q.EW(e, EW_HW + r * NWY + v) * Ld[DF_DW + v];
    for (int b = 0; b < NA; ++b) t += Q.EW(e, EW_HW + r * NWY + NW + b) * Ld[DF_DY + b];
    for (int a = 0; a < NX; ++a) t += Q.EW(e, EW_EWJ + a * NW + r) * Ld[DF_DNU + a];
    for (int q = 0; q < NE; ++q) t += Q.EW(e, EW_JDW + q * NW + r) * Ld[DF_DYD + q];
    Ld[DF_RHS + r] = -t;
  }
  T.gsync();
  for (int r = lane; r < NW; r += GS) {
    double t = 0.0;
    for (int c = 0; c < NW; ++c) t += Q.EW(e, EW_LU + c * NW + r) * Ld[DF_RHS + c];
    Q.dlam[row0 + r] = t;
  }
  T.gsync();
}
