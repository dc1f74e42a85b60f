// dompc_kernel.h - structured interior-point solver for the multi-stage (scenario tree) NLP of
// do-mpc's MPC.make_step(), written for gfx950 (MI355X).  One workgroup solves one problem from
// x0-in to u0-out: model evaluation + per-edge collocation condensing, tree Riccati recursion,
// fraction-to-boundary, filter line search and barrier update all stay on the device.
//
// What it replaces in the reference (everything CasADi/IPOPT/MUMPS do inside
// `r = self.S(**kwargs)`, /root/reference/do_mpc/optimizer.py:770):
//   nlp_f / nlp_g / nlp_grad_f / nlp_jac_g / nlp_hess_l   -> eval_edge()  (per edge, per collocation point)
//   MUMPS LDL^T of the sparse KKT matrix                  -> condense (LU of the collocation block) +
//                                                            riccati_backward()/forward() on the tree
//   IPOPT's filter line search / mu update / termination  -> solve_problem()
// The algorithm constants are IPOPT's defaults (Waechter & Biegler 2006), see include/dompc_ipm.h.
//
// The file is compiled twice from the same text:
//   * by hipcc --offload-arch=gfx950 into the per-model code object (product path), and
//   * by g++ with -DDOMPC_HOST_EMU into a test-only library where a "workgroup" is one host
//     thread (tests/ build it; the product never loads it).
// It must be included after the generated model header (do_mpc_amd/lowering.py).
#pragma once
#include <math.h>
#include <stdint.h>
#include "dompc_kargs.h"

namespace dompc {

constexpr int NX = DOMPC_NX, NU = DOMPC_NU, NP = DOMPC_NP, NTVP = DOMPC_NTVP;
constexpr int NE = DOMPC_NE, NS = DOMPC_NS;
constexpr int DEG = DOMPC_DEG, NI = DOMPC_NI, M = DOMPC_M;
constexpr int NA = NX + NU;          // (x,u) of a stage == augmented state (x,u_prev)
constexpr int NV = NU + NS;          // decision variables of a node: u then eps
constexpr int NYT = NA + NV;         // node quadratic: (x, u_prev, u, eps)
constexpr int NW = M * NX;           // collocation unknowns of an edge (incl. the xkf slot)
constexpr int NCOLL = NI * DEG;      // collocation points evaluated per edge
constexpr int RPE = NW + NX + NE;    // constraint rows per edge
constexpr int NE1 = NE > 0 ? NE : 1;
constexpr int NS1 = NS > 0 ? NS : 1;
constexpr int NW1 = NW > 0 ? NW : 1;
constexpr int MAX_FILTER = 48;
constexpr int RED_MAX = 12;          // values reduced per pass

// per-edge interleaved workspace (index [field + i][edge]) -------------------------------------
constexpr int EW_LU = 0;
constexpr int EW_PIV = EW_LU + NW * NW;
constexpr int EW_W = EW_PIV + NW;            // NW x NA, row-major
constexpr int EW_W0 = EW_W + NW * NA;
constexpr int EW_HP = EW_W0 + NW;            // NCOLL x NA x NA (lambda-weighted dyn Hessians)
constexpr int EW_SIGW = EW_HP + NCOLL * NA * NA;
constexpr int EW_RW = EW_SIGW + NW;
constexpr int EW_JD = EW_RW + NW;            // NE x NA
constexpr int EW_SIZE = EW_JD + NE * NA + 1;

// per-edge shared (contiguous per edge) --------------------------------------------------------
constexpr int ES_AB = 0;                     // NX x NA
constexpr int ES_CV = ES_AB + NX * NA;
constexpr int ES_QT = ES_CV + NX;            // NA x NA
constexpr int ES_QV = ES_QT + NA * NA;
constexpr int ES_WTW = ES_QV + NA;
constexpr int ES_WTW0 = ES_WTW + NA * NA;
constexpr int ES_RY = ES_WTW0 + NA;          // G_y' lam + sf*omega*grad l + Jd' yd      (NA)
constexpr int ES_GFY = ES_RY + NA;           // sf*omega*grad l                            (NA)
constexpr int ES_MG = ES_GFY + NA;           // sf*omega*grad m (last edges)               (NX)
constexpr int ES_MH = ES_MG + NX;            // sf*omega*hess m                            (NX x NX)
constexpr int ES_SIGS = ES_MH + NX * NX;     // NE
constexpr int ES_RDN = ES_SIGS + NE;         // d - s
constexpr int ES_RSN = ES_RDN + NE;          // -yd - mu/(s-sl) + mu/(su-s)
constexpr int ES_TP = ES_RSN + NE;           // NA x NA : P_c * Atilde (y columns)
constexpr int ES_TV = ES_TP + NA * NA;       // NA : P_c*ctilde + p_c
constexpr int ES_ACL = ES_TV + NA;          // NA x NA : Atilde * [I;K] (closed-loop map)
constexpr int ES_CCL = ES_ACL + NA * NA;     // NA : Atilde*[0;kv] + ctilde
constexpr int ES_OBJ = ES_CCL + NA;
constexpr int ES_SIZE = ES_OBJ + 1;

// per node -------------------------------------------------------------------------------------
constexpr int ND_P = 0;                      // NA x NA
constexpr int ND_PV = ND_P + NA * NA;
constexpr int ND_K = ND_PV + NA;             // NV x NA
constexpr int ND_KV = ND_K + NV * NA;
constexpr int ND_Q = ND_KV + NV;             // NYT x NYT
constexpr int ND_QV = ND_Q + NYT * NYT;
constexpr int ND_DXT = ND_QV + NYT;          // NA
constexpr int ND_L = ND_DXT + NA;            // NV x NV
constexpr int ND_QO = ND_L + NV * NV;        // NYT x NYT : node quadratic without the children's value functions
constexpr int ND_QOV = ND_QO + NYT * NYT;
constexpr int ND_SIZE = ND_QOV + NYT;

struct WsLayout {
  int64_t x, zl, zu, lb, ub, dx, gf, rd, xt, dzl, dzu;
  int64_t lam, dlam, c, ct;
  int64_t s, zsl, zsu, sl, su, ds, st, dzsl, dzsu;
  int64_t ew, es, nd, total;
};

DOMPC_HD inline WsLayout ws_layout(int n_opt_x, int n_g, int n_edges, int e_pad, int n_nodes) {
  WsLayout L;
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += (n + 7) & ~int64_t(7); return r; };
  L.x = take(n_opt_x); L.zl = take(n_opt_x); L.zu = take(n_opt_x); L.lb = take(n_opt_x); L.ub = take(n_opt_x);
  L.dx = take(n_opt_x); L.gf = take(n_opt_x); L.rd = take(n_opt_x); L.xt = take(n_opt_x);
  L.dzl = take(n_opt_x); L.dzu = take(n_opt_x);
  L.lam = take(n_g); L.dlam = take(n_g); L.c = take(n_g); L.ct = take(n_g);
  int64_t nsl = (int64_t)n_edges * NE1;
  L.s = take(nsl); L.zsl = take(nsl); L.zsu = take(nsl); L.sl = take(nsl); L.su = take(nsl);
  L.ds = take(nsl); L.st = take(nsl); L.dzsl = take(nsl); L.dzsu = take(nsl);
  L.ew = take((int64_t)EW_SIZE * e_pad);
  L.es = take((int64_t)ES_SIZE * n_edges);
  L.nd = take((int64_t)ND_SIZE * n_nodes);
  L.total = o;
  return L;
}

// ------------------------------------------------------------------------------------------------
// workgroup context
struct Thr {
  int tid, nt;
  double* red;      // LDS: RED_MAX * nt doubles
  double* filt;     // LDS: 2*MAX_FILTER doubles
  int* flags;       // LDS: 8 ints
  DOMPC_DEV void sync() const {
#ifndef DOMPC_HOST_EMU
    __syncthreads();
#endif
  }
};

enum RedOp { R_SUM = 0, R_MAX = 1, R_MIN = 2 };

// Reduce n values per thread across the workgroup; result broadcast to every thread.
template <int N_>
DOMPC_DEV void wg_reduce(const Thr& T, double (&v)[N_], const int (&op)[N_]) {
  static_assert(N_ <= RED_MAX, "too many values");
  if (T.nt == 1) return;
  for (int i = 0; i < N_; ++i) T.red[i * T.nt + T.tid] = v[i];
  T.sync();
  for (int s = T.nt >> 1; s > 0; s >>= 1) {
    if (T.tid < s) {
      for (int i = 0; i < N_; ++i) {
        double a = T.red[i * T.nt + T.tid], b = T.red[i * T.nt + T.tid + s];
        T.red[i * T.nt + T.tid] = op[i] == R_SUM ? a + b : (op[i] == R_MAX ? fmax(a, b) : fmin(a, b));
      }
    }
    T.sync();
  }
  for (int i = 0; i < N_; ++i) v[i] = T.red[i * T.nt];
  T.sync();
}

// ------------------------------------------------------------------------------------------------
// per-problem view
struct Prob {
  const KArgs* A;
  const double* P;                                   // opt_p of this problem
  double *x, *zl, *zu, *lb, *ub, *dx, *gf, *rd, *xt, *dzl, *dzu;
  double *lam, *dlam, *c, *ct;
  double *s, *zsl, *zsu, *sl, *su, *ds, *st, *dzsl, *dzsu;
  double *ew, *es, *nd;
  int e_pad;
  double sf;                                         // objective scaling
  double mu;
  DOMPC_DEV double& EW(int e, int i) const { return ew[(int64_t)i * e_pad + e]; }
  DOMPC_DEV double* ES(int e) const { return es + (int64_t)e * ES_SIZE; }
  DOMPC_DEV double* ND(int n) const { return nd + (int64_t)n * ND_SIZE; }
};

DOMPC_DEV inline Prob make_prob(const KArgs& A, int slot, const double* P) {
  WsLayout L = ws_layout(A.n_opt_x, A.n_g, A.n_edges, A.e_pad, A.n_nodes);
  double* w = A.ws + (int64_t)slot * A.ws_stride;
  Prob p;
  p.A = &A; p.P = P;
  p.x = w + L.x; p.zl = w + L.zl; p.zu = w + L.zu; p.lb = w + L.lb; p.ub = w + L.ub; p.dx = w + L.dx;
  p.gf = w + L.gf; p.rd = w + L.rd; p.xt = w + L.xt; p.dzl = w + L.dzl; p.dzu = w + L.dzu;
  p.lam = w + L.lam; p.dlam = w + L.dlam; p.c = w + L.c; p.ct = w + L.ct;
  p.s = w + L.s; p.zsl = w + L.zsl; p.zsu = w + L.zsu; p.sl = w + L.sl; p.su = w + L.su;
  p.ds = w + L.ds; p.st = w + L.st; p.dzsl = w + L.dzsl; p.dzsu = w + L.dzsu;
  p.ew = w + L.ew; p.es = w + L.es; p.nd = w + L.nd;
  p.e_pad = A.e_pad; p.sf = 1.0; p.mu = 0.0;
  return p;
}

// slot of collocation point r of finite element i inside the edge's w block (optimizer.py:905-935)
DOMPC_DEV constexpr int slot_of(int i, int r) { return i == 0 ? r - 1 : DEG + (i - 1) * (DEG + 1) + r; }
DOMPC_DEV constexpr int next_slot(int i) { return (i + 1 < NI) ? slot_of(i + 1, 0) : M - 1; }

DOMPC_DEV inline double bar_grad(double x, double l, double u, double mu) {
  double g = 0.0;
  if (l > -INFINITY) g -= mu / (x - l);
  if (u < INFINITY) g += mu / (u - x);
  return g;
}
DOMPC_DEV inline double sigma_of(double x, double l, double u, double zl, double zu) {
  double sg = 0.0;
  if (l > -INFINITY) sg += zl / (x - l);
  if (u < INFINITY) sg += zu / (u - x);
  return sg;
}

// ================================================================================================
// Trial evaluation: constraint residuals + objective share of one edge at `xv` (no derivatives).
// nlp_g / nlp_f of the reference for the rows/terms owned by edge e.
DOMPC_DEV inline double eval_edge_f(const Prob& Q, int e, const double* xv, const double* sv, double* cv) {
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
  if (M == 0) {
    dompc_dyn_f(xn, un, tvp, pp, f);
    for (int a = 0; a < NX; ++a) cv[row0 + a] = f[a] - xc[a];
  } else {
    for (int i = 0; i < NI; ++i) {
      const double* xi0 = (i == 0) ? xn : w + slot_of(i, 0) * NX;
      const int rb = row0 + i * (DEG + 1) * NX;
      for (int j = 1; j <= DEG; ++j) {
        const double* xij = w + slot_of(i, j) * NX;
        dompc_dyn_f(xij, un, tvp, pp, f);
        for (int a = 0; a < NX; ++a) {
          double xp = DOMPC_C[0 * (DEG + 1) + j] * xi0[a];
          for (int r = 1; r <= DEG; ++r) xp += DOMPC_C[r * (DEG + 1) + j] * w[slot_of(i, r) * NX + a];
          cv[rb + (j - 1) * NX + a] = f[a] - xp;
        }
      }
      const double* xnext = w + next_slot(i) * NX;
      for (int a = 0; a < NX; ++a) {
        double xf = DOMPC_D[0] * xi0[a];
        for (int r = 1; r <= DEG; ++r) xf += DOMPC_D[r] * w[slot_of(i, r) * NX + a];
        cv[rb + DEG * NX + a] = xnext[a] - xf;
      }
    }
    for (int a = 0; a < NX; ++a) cv[row0 + NW + a] = w[(M - 1) * NX + a] - xc[a];
  }
  double obj = om * dompc_lterm_f(xn, un, tvp, pp);
  if (k == A.N - 1) obj += om * dompc_mterm_f(xc, Q.P + A.p_off_tvp + (k + 1) * NTVP, pp);
  if (NE > 0) {
    double d[NE1];
    dompc_nlcons_f(xn, un, tvp, pp, d);
    const double* eps = (NS > 0) ? xv + A.node_eps_off[n] : nullptr;
    for (int i = 0; i < NE; ++i) {
      if (DOMPC_NL_SLACK[i] >= 0) d[i] -= eps[DOMPC_NL_SLACK[i]];
      cv[row0 + NW + NX + i] = d[i] - sv[e * NE1 + i];
    }
    for (int q = 0; q < NS; ++q) obj += Q.sf * DOMPC_EPS_PEN[q] * eps[q];
  }
  return obj;
}

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
  if (A.node_u_off[n] < 0) return 0.0;
  double tmp[NU];
  const double* up = uprev_ptr(Q, n, xv, tmp);
  const double* u = xv + A.node_u_off[n];
  const double rw = node_rweight(Q, n);
  double v = 0.0;
  for (int i = 0; i < NU; ++i) v += rw * DOMPC_RTERM[i] * (u[i] - up[i]) * (u[i] - up[i]);
  return v;
}

// ================================================================================================
// Derivative evaluation + condensing of one edge (thread-per-edge).
//  builds G_w, G_y, residuals; LU of G_w; W = -G_w^-1 G_y, w0 = -G_w^-1 r_g; [A|B] = S W, c;
//  condensed Hessian/gradient share over y = (x_n,u_n); w-part of the dual residual.
// `ls_mode`: 0 normal Newton system.
DOMPC_DEV inline int eval_edge(const Prob& Q, int e, double mu) {
  const KArgs& A = *Q.A;
  const int n = A.edge_parent[e], cn = A.edge_child[e], k = A.edge_level[e];
  const double* xn = Q.x + A.node_x_off[n];
  const double* un = Q.x + A.node_u_off[n];
  const double* xc = Q.x + A.node_x_off[cn];
  const int woff = A.edge_w_off[e];
  const double* w = Q.x + woff;
  const double* pp = Q.P + A.p_off_p + A.edge_pidx[e] * NP;
  const double* tvp = Q.P + A.p_off_tvp + k * NTVP;
  const int row0 = A.edge_row0[e];
  const double om = A.edge_omega[e] * Q.sf;
  const double* lam_e = Q.lam + row0;
  const double* nu_e = Q.lam + row0 + NW;
  double* S_ = Q.ES(e);
  int fail = 0;

  double f[NX], J[NX * NA], H[NA * NA];
  double ry[NA];
  for (int a = 0; a < NA; ++a) ry[a] = 0.0;
  for (int a = 0; a < NA * NA; ++a) S_[ES_QT + a] = 0.0;

  if (M == 0) {
    // discrete model: x_c = f(x_n,u_n); rows: f - x_c (multiplier nu_e)
    dompc_dyn(xn, un, tvp, pp, nu_e, f, J, H);
    for (int a = 0; a < NX; ++a) {
      Q.c[row0 + a] = f[a] - xc[a];
      S_[ES_CV + a] = f[a] - xc[a];
      for (int b = 0; b < NA; ++b) S_[ES_AB + a * NA + b] = J[a * NA + b];
    }
    for (int b = 0; b < NA; ++b) {
      double t = 0.0;
      for (int a = 0; a < NX; ++a) t += J[a * NA + b] * nu_e[a];
      ry[b] = t;
    }
    for (int a = 0; a < NA * NA; ++a) { S_[ES_QT + a] = H[a]; S_[ES_WTW + a] = 0.0; }
    for (int a = 0; a < NA; ++a) { S_[ES_QV + a] = 0.0; S_[ES_WTW0 + a] = 0.0; }
  } else {
    // ---- zero G_w (-> EW_LU), G_y|r_g (-> EW_W, EW_W0)
    for (int i = 0; i < NW * NW; ++i) Q.EW(e, EW_LU + i) = 0.0;
    for (int i = 0; i < NW * NA; ++i) Q.EW(e, EW_W + i) = 0.0;
    double huu[NU * NU];
    for (int i = 0; i < NU * NU; ++i) huu[i] = 0.0;
    for (int i = 0; i < NI; ++i) {
      const double* xi0 = (i == 0) ? xn : w + slot_of(i, 0) * NX;
      const int rbl = i * (DEG + 1) * NX;    // local row base
      for (int j = 1; j <= DEG; ++j) {
        const int sl = slot_of(i, j);
        const double* xij = w + sl * NX;
        const int rb = rbl + (j - 1) * NX;
        const int pt = i * DEG + (j - 1);
        dompc_dyn(xij, un, tvp, pp, lam_e + rb, f, J, H);
        for (int a = 0; a < NA * NA; ++a) Q.EW(e, EW_HP + pt * NA * NA + a) = H[a];
        for (int a = 0; a < NU; ++a)
          for (int b = 0; b < NU; ++b) huu[a * NU + b] += H[(NX + a) * NA + NX + b];
        for (int a = 0; a < NX; ++a) {
          double xp = DOMPC_C[0 * (DEG + 1) + j] * xi0[a];
          for (int r = 1; r <= DEG; ++r) xp += DOMPC_C[r * (DEG + 1) + j] * w[slot_of(i, r) * NX + a];
          const double res = f[a] - xp;
          Q.c[row0 + rb + a] = res;
          Q.EW(e, EW_W0 + rb + a) = res;
          for (int b = 0; b < NX; ++b) Q.EW(e, EW_LU + (rb + a) * NW + sl * NX + b) += J[a * NA + b];
          for (int b = 0; b < NU; ++b) Q.EW(e, EW_W + (rb + a) * NA + NX + b) = J[a * NA + NX + b];
          for (int r = 0; r <= DEG; ++r) {
            const double cr = DOMPC_C[r * (DEG + 1) + j];
            if (i == 0 && r == 0) Q.EW(e, EW_W + (rb + a) * NA + a) -= cr;
            else Q.EW(e, EW_LU + (rb + a) * NW + slot_of(i, r) * NX + a) -= cr;
          }
        }
      }
      const int rb = rbl + DEG * NX;
      const int ns_ = next_slot(i);
      for (int a = 0; a < NX; ++a) {
        double xf = DOMPC_D[0] * xi0[a];
        for (int r = 1; r <= DEG; ++r) xf += DOMPC_D[r] * w[slot_of(i, r) * NX + a];
        const double res = w[ns_ * NX + a] - xf;
        Q.c[row0 + rb + a] = res;
        Q.EW(e, EW_W0 + rb + a) = res;
        Q.EW(e, EW_LU + (rb + a) * NW + ns_ * NX + a) += 1.0;
        for (int r = 0; r <= DEG; ++r) {
          if (i == 0 && r == 0) Q.EW(e, EW_W + (rb + a) * NA + a) -= DOMPC_D[0];
          else Q.EW(e, EW_LU + (rb + a) * NW + slot_of(i, r) * NX + a) -= DOMPC_D[r];
        }
      }
    }
    // continuity to the child node: xkf - x_c
    double rc[NX];
    for (int a = 0; a < NX; ++a) {
      rc[a] = w[(M - 1) * NX + a] - xc[a];
      Q.c[row0 + NW + a] = rc[a];
    }
    // ---- dual residual pieces that need G_w / G_y before they are overwritten
    for (int col = 0; col < NW; ++col) {
      double t = 0.0;
      for (int r = 0; r < NW; ++r) t += Q.EW(e, EW_LU + r * NW + col) * lam_e[r];
      if (col >= (M - 1) * NX) t += nu_e[col - (M - 1) * NX];
      const int gi = woff + col;
      const double xv = Q.x[gi], l = Q.lb[gi], u = Q.ub[gi];
      Q.gf[gi] = 0.0;
      Q.rd[gi] = t - Q.zl[gi] + Q.zu[gi];
      Q.EW(e, EW_RW + col) = t + bar_grad(xv, l, u, mu);
      Q.EW(e, EW_SIGW + col) = sigma_of(xv, l, u, Q.zl[gi], Q.zu[gi]);
    }
    for (int b = 0; b < NA; ++b) {
      double t = 0.0;
      for (int r = 0; r < NW; ++r) t += Q.EW(e, EW_W + r * NA + b) * lam_e[r];
      ry[b] = t;
    }
    // ---- LU with partial pivoting (in place, interleaved storage)
    for (int kk = 0; kk < NW; ++kk) {
      int pv = kk;
      double best = fabs(Q.EW(e, EW_LU + kk * NW + kk));
      for (int r = kk + 1; r < NW; ++r) {
        const double v = fabs(Q.EW(e, EW_LU + r * NW + kk));
        if (v > best) { best = v; pv = r; }
      }
      Q.EW(e, EW_PIV + kk) = (double)pv;
      if (!(best > 1e-300)) { fail = 1; best = 1.0; Q.EW(e, EW_LU + pv * NW + kk) = 1.0; }
      if (pv != kk) {
        for (int cix = 0; cix < NW; ++cix) {
          const double t = Q.EW(e, EW_LU + kk * NW + cix);
          Q.EW(e, EW_LU + kk * NW + cix) = Q.EW(e, EW_LU + pv * NW + cix);
          Q.EW(e, EW_LU + pv * NW + cix) = t;
        }
        for (int cix = 0; cix < NA; ++cix) {
          const double t = Q.EW(e, EW_W + kk * NA + cix);
          Q.EW(e, EW_W + kk * NA + cix) = Q.EW(e, EW_W + pv * NA + cix);
          Q.EW(e, EW_W + pv * NA + cix) = t;
        }
        const double t = Q.EW(e, EW_W0 + kk);
        Q.EW(e, EW_W0 + kk) = Q.EW(e, EW_W0 + pv);
        Q.EW(e, EW_W0 + pv) = t;
      }
      const double inv = 1.0 / Q.EW(e, EW_LU + kk * NW + kk);
      for (int r = kk + 1; r < NW; ++r) {
        const double lf = Q.EW(e, EW_LU + r * NW + kk) * inv;
        if (lf != 0.0) {
          Q.EW(e, EW_LU + r * NW + kk) = lf;
          for (int cix = kk + 1; cix < NW; ++cix)
            Q.EW(e, EW_LU + r * NW + cix) -= lf * Q.EW(e, EW_LU + kk * NW + cix);
          // forward elimination of the right-hand sides at the same time
          for (int cix = 0; cix < NA; ++cix) Q.EW(e, EW_W + r * NA + cix) -= lf * Q.EW(e, EW_W + kk * NA + cix);
          Q.EW(e, EW_W0 + r) -= lf * Q.EW(e, EW_W0 + kk);
        } else {
          Q.EW(e, EW_LU + r * NW + kk) = 0.0;
        }
      }
    }
    // back substitution, then negate:  W = -G_w^-1 G_y,  w0 = -G_w^-1 r_g
    for (int r = NW - 1; r >= 0; --r) {
      const double inv = 1.0 / Q.EW(e, EW_LU + r * NW + r);
      for (int cix = 0; cix < NA; ++cix) {
        double t = Q.EW(e, EW_W + r * NA + cix);
        for (int q = r + 1; q < NW; ++q) t -= Q.EW(e, EW_LU + r * NW + q) * Q.EW(e, EW_W + q * NA + cix);
        Q.EW(e, EW_W + r * NA + cix) = t * inv;
      }
      double t = Q.EW(e, EW_W0 + r);
      for (int q = r + 1; q < NW; ++q) t -= Q.EW(e, EW_LU + r * NW + q) * Q.EW(e, EW_W0 + q);
      Q.EW(e, EW_W0 + r) = t * inv;
    }
    for (int i = 0; i < NW * NA; ++i) Q.EW(e, EW_W + i) = -Q.EW(e, EW_W + i);
    for (int i = 0; i < NW; ++i) Q.EW(e, EW_W0 + i) = -Q.EW(e, EW_W0 + i);
    // [A|B] = S W ; c = S w0 + r_c
    for (int a = 0; a < NX; ++a) {
      for (int b = 0; b < NA; ++b) S_[ES_AB + a * NA + b] = Q.EW(e, EW_W + ((M - 1) * NX + a) * NA + b);
      S_[ES_CV + a] = Q.EW(e, EW_W0 + (M - 1) * NX + a) + rc[a];
    }
    // ---- condensed Hessian  Qt = Huu-part + Hyw W + (Hyw W)' + W' Hww W ;  Qv = Hyw w0 + W'(rw + Hww w0)
    //      WTW = W'W, WTW0 = W'w0 (for the delta_w regularisation)
    for (int a = 0; a < NA * NA; ++a) S_[ES_WTW + a] = 0.0;
    for (int a = 0; a < NA; ++a) { S_[ES_QV + a] = 0.0; S_[ES_WTW0 + a] = 0.0; }
    for (int a = 0; a < NU; ++a)
      for (int b = 0; b < NU; ++b) S_[ES_QT + (NX + a) * NA + NX + b] += huu[a * NU + b];
    // row-by-row over w: t1 = (Hww W)[row,:], t0 = (Hww w0)[row]
    for (int row = 0; row < NW; ++row) {
      double t1[NA];
      const double sg = Q.EW(e, EW_SIGW + row);
      for (int b = 0; b < NA; ++b) t1[b] = sg * Q.EW(e, EW_W + row * NA + b);
      double t0 = sg * Q.EW(e, EW_W0 + row);
      // which collocation point owns this row's slot?
      const int sl = row / NX, a = row % NX;
      int pt = -1;
      for (int i = 0; i < NI; ++i)
        for (int j = 1; j <= DEG; ++j)
          if (slot_of(i, j) == sl) pt = i * DEG + (j - 1);
      if (pt >= 0) {
        for (int a2 = 0; a2 < NX; ++a2) {
          const double h = Q.EW(e, EW_HP + pt * NA * NA + a * NA + a2);
          if (h != 0.0) {
            for (int b = 0; b < NA; ++b) t1[b] += h * Q.EW(e, EW_W + (sl * NX + a2) * NA + b);
            t0 += h * Q.EW(e, EW_W0 + sl * NX + a2);
          }
        }
        // Hwu contributions: Qt[u,:] += Hux W[row,:] ; Qt[:,u] += same' ; Qv[u] += Hux w0[row]
        for (int ub = 0; ub < NU; ++ub) {
          const double h = Q.EW(e, EW_HP + pt * NA * NA + a * NA + NX + ub);   // H[x_a][u_ub]
          if (h != 0.0) {
            for (int b = 0; b < NA; ++b) {
              const double wv = Q.EW(e, EW_W + row * NA + b);
              S_[ES_QT + (NX + ub) * NA + b] += h * wv;
              S_[ES_QT + b * NA + NX + ub] += h * wv;
            }
            S_[ES_QV + NX + ub] += h * Q.EW(e, EW_W0 + row);
            // and W' (Hwu du-part) is covered by the symmetric term above; gradient part via rw below
          }
        }
      }
      const double rwv = Q.EW(e, EW_RW + row) + t0;
      for (int a1 = 0; a1 < NA; ++a1) {
        const double wa = Q.EW(e, EW_W + row * NA + a1);
        if (wa != 0.0) {
          for (int b = 0; b < NA; ++b) {
            S_[ES_QT + a1 * NA + b] += wa * t1[b];
            S_[ES_WTW + a1 * NA + b] += wa * Q.EW(e, EW_W + row * NA + b);
          }
          S_[ES_QV + a1] += wa * rwv;
          S_[ES_WTW0 + a1] += wa * Q.EW(e, EW_W0 + row);
        }
      }
    }
  }
  // ---- stage cost (weight sf*omega), nl_cons
  double lval, gl[NA];
  dompc_lterm(xn, un, tvp, pp, &lval, gl, H);
  double obj = om * lval;
  for (int a = 0; a < NA; ++a) {
    S_[ES_GFY + a] = om * gl[a];
    ry[a] += om * gl[a];
    for (int b = 0; b < NA; ++b) S_[ES_QT + a * NA + b] += om * H[a * NA + b];
  }
  if (k == A.N - 1) {
    double mval, gm[NX], Hm[NX * NX];
    dompc_mterm(xc, Q.P + A.p_off_tvp + (k + 1) * NTVP, pp, &mval, gm, Hm);
    obj += om * mval;
    for (int a = 0; a < NX; ++a) S_[ES_MG + a] = om * gm[a];
    for (int a = 0; a < NX * NX; ++a) S_[ES_MH + a] = om * Hm[a];
  }
  if (NE > 0) {
    const double* yd = Q.lam + row0 + NW + NX;
    double d[NE1], Jd[NE1 * NA];
    dompc_nlcons(xn, un, tvp, pp, yd, d, Jd, H);
    const double* eps = (NS > 0) ? Q.x + A.node_eps_off[n] : nullptr;
    for (int a = 0; a < NA * NA; ++a) S_[ES_QT + a] += H[a];
    for (int i = 0; i < NE; ++i) {
      if (DOMPC_NL_SLACK[i] >= 0) d[i] -= eps[DOMPC_NL_SLACK[i]];
      const int si = e * NE1 + i;
      const double sv = Q.s[si], l = Q.sl[si], u = Q.su[si];
      Q.c[row0 + NW + NX + i] = d[i] - sv;
      S_[ES_RDN + i] = d[i] - sv;
      S_[ES_SIGS + i] = sigma_of(sv, l, u, Q.zsl[si], Q.zsu[si]);
      S_[ES_RSN + i] = -yd[i] + bar_grad(sv, l, u, mu);
      for (int b = 0; b < NA; ++b) {
        Q.EW(e, EW_JD + i * NA + b) = Jd[i * NA + b];
        ry[b] += Jd[i * NA + b] * yd[i];
      }
    }
    for (int q = 0; q < NS; ++q) obj += Q.sf * DOMPC_EPS_PEN[q] * eps[q];
  }
  for (int a = 0; a < NA; ++a) S_[ES_RY + a] = ry[a];
  S_[ES_OBJ] = obj;
  return fail;
}

// ================================================================================================
// Gradient / dual-residual assembly for the variables owned by node n (x_n, u_n, eps_n).
DOMPC_DEV inline void assemble_node(const Prob& Q, int n) {
  const KArgs& A = *Q.A;
  const int cs = A.node_child_start[n], cc = A.node_child_count[n];
  const int xo = A.node_x_off[n];
  double gx[NX], rx[NX];
  for (int a = 0; a < NX; ++a) { gx[a] = 0.0; rx[a] = 0.0; }
  for (int j = 0; j < cc; ++j) {
    const double* S_ = Q.ES(cs + j);
    for (int a = 0; a < NX; ++a) { gx[a] += S_[ES_GFY + a]; rx[a] += S_[ES_RY + a]; }
  }
  const int ie = A.node_in_edge[n];
  if (ie >= 0) {
    const double* nu_in = Q.lam + A.edge_row0[ie] + NW;
    for (int a = 0; a < NX; ++a) rx[a] -= nu_in[a];
    if (cc == 0) {
      const double* S_ = Q.ES(ie);
      for (int a = 0; a < NX; ++a) { gx[a] += S_[ES_MG + a]; rx[a] += S_[ES_MG + a]; }
    }
  } else {
    for (int a = 0; a < NX; ++a) rx[a] += Q.lam[a];
  }
  for (int a = 0; a < NX; ++a) {
    Q.gf[xo + a] = gx[a];
    Q.rd[xo + a] = rx[a] - Q.zl[xo + a] + Q.zu[xo + a];
  }
  if (cc == 0) return;
  const int uo = A.node_u_off[n];
  double tmp[NU];
  const double* up = uprev_ptr(Q, n, Q.x, tmp);
  const double rw = node_rweight(Q, n);
  for (int i = 0; i < NU; ++i) {
    double g = 0.0, r = 0.0;
    for (int j = 0; j < cc; ++j) {
      const double* S_ = Q.ES(cs + j);
      g += S_[ES_GFY + NX + i];
      r += S_[ES_RY + NX + i];
    }
    double rt = 2.0 * rw * DOMPC_RTERM[i] * (Q.x[uo + i] - up[i]);
    for (int j = 0; j < cc; ++j) {                 // children's rterm w.r.t. their u_prev = u_n
      const int cn = A.edge_child[cs + j];
      if (A.node_u_off[cn] >= 0) {
        const double rwc = node_rweight(Q, cn);
        rt -= 2.0 * rwc * DOMPC_RTERM[i] * (Q.x[A.node_u_off[cn] + i] - Q.x[uo + i]);
      }
    }
    Q.gf[uo + i] = g + rt;
    Q.rd[uo + i] = r + rt - Q.zl[uo + i] + Q.zu[uo + i];
  }
  if (NS > 0) {
    const int eo = A.node_eps_off[n];
    for (int q = 0; q < NS; ++q) {
      double g = cc * Q.sf * DOMPC_EPS_PEN[q];
      double r = g;
      for (int j = 0; j < cc; ++j) {
        const double* yd = Q.lam + A.edge_row0[cs + j] + NW + NX;
        for (int i = 0; i < NE; ++i)
          if (DOMPC_NL_SLACK[i] == q) r -= yd[i];
      }
      Q.gf[eo + q] = g;
      Q.rd[eo + q] = r - Q.zl[eo + q] + Q.zu[eo + q];
    }
  }
}

// ================================================================================================
// Tree Riccati recursion.  Value function of node n over its augmented state (x_n, u_prev_n):
//   V_n(d) = 1/2 d'P_n d + p_n'd   (Newton form: p built from dual residuals).
// Children are summed at branching nodes (non-anticipativity = shared variables, _mpc.py:1212-1216).
DOMPC_DEV inline int ycol(int yj) { return yj < NX ? yj : NA + (yj - NX); }

DOMPC_DEV inline int riccati_backward(const Thr& T, const Prob& Q, double mu, double delta) {
  const KArgs& A = *Q.A;
  if (T.tid == 0) T.flags[0] = 0;
  T.sync();
  for (int k = A.N; k >= 0; --k) {
    const int n0 = A.level_node_start[k], n1 = A.level_node_start[k + 1];
    const int nn = n1 - n0;
    if (k == A.N) {
      // leaves: P = sf*omega*Hm + Sigma_x, p = sf*omega*gm - nu_in + barrier
      for (int it = T.tid; it < nn * NA * (NA + 1); it += T.nt) {
        const int n = n0 + it / (NA * (NA + 1));
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
            v = S_[ES_MG + i] - Q.lam[A.edge_row0[ie] + NW + i] + bar_grad(Q.x[xo + i], Q.lb[xo + i], Q.ub[xo + i], mu);
          Nd[ND_PV + i] = v;
        }
      }
      T.sync();
      continue;
    }
    const int e0 = A.node_child_start[n0];
    const int e1 = A.node_child_start[n1 - 1] + A.node_child_count[n1 - 1];
    const int ne_ = e1 - e0;
    // (a) per child edge: TP = P_c * Atilde (y columns), TV = P_c*ctilde + p_c
    for (int it = T.tid; it < ne_ * NA * (NA + 1); it += T.nt) {
      const int e = e0 + it / (NA * (NA + 1));
      const int r = it % (NA * (NA + 1));
      const int i = r / (NA + 1), yj = r % (NA + 1);
      double* S_ = Q.ES(e);
      const double* Pc = Q.ND(A.edge_child[e]) + ND_P;
      if (yj < NA) {
        double t = 0.0;
        for (int a = 0; a < NX; ++a) t += Pc[i * NA + a] * S_[ES_AB + a * NA + yj];
        if (yj >= NX) t += Pc[i * NA + NX + (yj - NX)];
        S_[ES_TP + i * NA + yj] = t;
      } else {
        double t = Q.ND(A.edge_child[e])[ND_PV + i];
        for (int a = 0; a < NX; ++a) t += Pc[i * NA + a] * S_[ES_CV + a];
        S_[ES_TV + i] = t;
      }
    }
    T.sync();
    // (b) node quadratic Q (NYT x NYT) and q (NYT)
    for (int it = T.tid; it < nn * NYT * (NYT + 1); it += T.nt) {
      const int n = n0 + it / (NYT * (NYT + 1));
      const int r = it % (NYT * (NYT + 1));
      const int i = r / (NYT + 1), j = r % (NYT + 1);
      double* Nd = Q.ND(n);
      const int cs = A.node_child_start[n], cc = A.node_child_count[n];
      const int xo = A.node_x_off[n], uo = A.node_u_off[n];
      const int eo = NS > 0 ? A.node_eps_off[n] : -1;
      const double rw = node_rweight(Q, n);
      // classify index i (and j)
      // 0..NX-1: x ; NX..NA-1: u_prev ; NA..NA+NU-1: u ; NA+NU.. : eps
      const int yi = (i < NX) ? i : ((i >= NA && i < NA + NU) ? NX + (i - NA) : -1);   // y index or -1
      if (j < NYT) {
        const int yj2 = (j < NX) ? j : ((j >= NA && j < NA + NU) ? NX + (j - NA) : -1);
        double v = 0.0, vc = 0.0;
        if (i == j) {
          if (i < NX) v += sigma_of(Q.x[xo + i], Q.lb[xo + i], Q.ub[xo + i], Q.zl[xo + i], Q.zu[xo + i]) + delta;
          else if (i < NA) v += 2.0 * rw * DOMPC_RTERM[i - NX];
          else if (i < NA + NU) {
            const int g = uo + (i - NA);
            v += 2.0 * rw * DOMPC_RTERM[i - NA] + sigma_of(Q.x[g], Q.lb[g], Q.ub[g], Q.zl[g], Q.zu[g]) + delta;
          } else {
            const int g = eo + (i - NA - NU);
            v += sigma_of(Q.x[g], Q.lb[g], Q.ub[g], Q.zl[g], Q.zu[g]) + delta;
          }
        } else if (i >= NX && i < NA && j == i + NU) v -= 2.0 * rw * DOMPC_RTERM[i - NX];
        else if (j >= NX && j < NA && i == j + NU) v -= 2.0 * rw * DOMPC_RTERM[j - NX];
        for (int c = 0; c < cc; ++c) {
          const int e = cs + c;
          const double* S_ = Q.ES(e);
          if (yi >= 0 && yj2 >= 0) {
            v += S_[ES_QT + yi * NA + yj2] + delta * S_[ES_WTW + yi * NA + yj2];
            double t = 0.0;
            for (int a = 0; a < NX; ++a) t += S_[ES_AB + a * NA + yi] * S_[ES_TP + a * NA + yj2];
            if (yi >= NX) t += S_[ES_TP + (NX + yi - NX) * NA + yj2];
            vc += t;
          }
          if (NE > 0) {
            for (int q = 0; q < NE; ++q) {
              const double sg = S_[ES_SIGS + q] + delta;
              double ji = 0.0, jj = 0.0;
              if (yi >= 0) ji = Q.EW(e, EW_JD + q * NA + yi);
              else if (i >= NA + NU && DOMPC_NL_SLACK[q] == i - NA - NU) ji = -1.0;
              if (yj2 >= 0) jj = Q.EW(e, EW_JD + q * NA + yj2);
              else if (j >= NA + NU && DOMPC_NL_SLACK[q] == j - NA - NU) jj = -1.0;
              v += sg * ji * jj;
            }
          }
        }
        Nd[ND_Q + i * NYT + j] = v + vc;
        Nd[ND_QO + i * NYT + j] = v;
      } else {
        double v = 0.0, vc = 0.0;
        double tmp[NU];
        if (i < NX) {
          const int ie = A.node_in_edge[n];
          v += (ie >= 0) ? -Q.lam[A.edge_row0[ie] + NW + i] : Q.lam[i];
          v += bar_grad(Q.x[xo + i], Q.lb[xo + i], Q.ub[xo + i], mu);
        } else if (i < NA) {
          const double* up = uprev_ptr(Q, n, Q.x, tmp);
          v -= 2.0 * rw * DOMPC_RTERM[i - NX] * (Q.x[uo + i - NX] - up[i - NX]);
        } else if (i < NA + NU) {
          const double* up = uprev_ptr(Q, n, Q.x, tmp);
          const int g = uo + (i - NA);
          v += 2.0 * rw * DOMPC_RTERM[i - NA] * (Q.x[g] - up[i - NA]) + bar_grad(Q.x[g], Q.lb[g], Q.ub[g], mu);
        } else {
          const int q = i - NA - NU;
          const int g = eo + q;
          v += cc * Q.sf * DOMPC_EPS_PEN[q] + bar_grad(Q.x[g], Q.lb[g], Q.ub[g], mu);
        }
        for (int c = 0; c < cc; ++c) {
          const int e = cs + c;
          const double* S_ = Q.ES(e);
          if (yi >= 0) {
            v += S_[ES_RY + yi] + S_[ES_QV + yi] + delta * S_[ES_WTW0 + yi];
            double t = 0.0;
            for (int a = 0; a < NX; ++a) t += S_[ES_AB + a * NA + yi] * S_[ES_TV + a];
            if (yi >= NX) t += S_[ES_TV + yi];
            vc += t;
          }
          if (NE > 0) {
            const double* yd = Q.lam + A.edge_row0[e] + NW + NX;
            for (int q = 0; q < NE; ++q) {
              const double sg = S_[ES_SIGS + q] + delta;
              double ji = 0.0;
              if (yi >= 0) ji = Q.EW(e, EW_JD + q * NA + yi);
              else if (i >= NA + NU && DOMPC_NL_SLACK[q] == i - NA - NU) { ji = -1.0; v -= yd[q]; }
              v += ji * (sg * S_[ES_RDN + q] + S_[ES_RSN + q]);
            }
          }
        }
        Nd[ND_QV + i] = v + vc;
        Nd[ND_QOV + i] = v;
      }
    }
    T.sync();
    // (c) Cholesky of Qvv (thread per node)
    for (int it = T.tid; it < nn; it += T.nt) {
      double* Nd = Q.ND(n0 + it);
      double L[NV * NV];
      int bad = 0;
      for (int i = 0; i < NV; ++i)
        for (int j = 0; j <= i; ++j) {
          double t = Nd[ND_Q + (NA + i) * NYT + NA + j];
          for (int q = 0; q < j; ++q) t -= L[i * NV + q] * L[j * NV + q];
          if (i == j) {
            if (!(t > 0.0)) { bad = 1; t = 1.0; }
            L[i * NV + i] = sqrt(t);
          } else {
            L[i * NV + j] = t / L[j * NV + j];
          }
        }
      for (int i = 0; i < NV; ++i)
        for (int j = 0; j <= i; ++j) Nd[ND_L + i * NV + j] = L[i * NV + j];
      if (bad) T.flags[0] = 1;
    }
    T.sync();
    if (T.flags[0]) return 1;
    // (d) K = -Qvv^-1 Qvx, kv = -Qvv^-1 qv  (thread per (node, column))
    for (int it = T.tid; it < nn * (NA + 1); it += T.nt) {
      double* Nd = Q.ND(n0 + it / (NA + 1));
      const int j = it % (NA + 1);
      double y[NV];
      for (int i = 0; i < NV; ++i) {
        double t = (j < NA) ? Nd[ND_Q + (NA + i) * NYT + j] : Nd[ND_QV + NA + i];
        for (int q = 0; q < i; ++q) t -= Nd[ND_L + i * NV + q] * y[q];
        y[i] = t / Nd[ND_L + i * NV + i];
      }
      for (int i = NV - 1; i >= 0; --i) {
        double t = y[i];
        for (int q = i + 1; q < NV; ++q) t -= Nd[ND_L + q * NV + i] * y[q];
        y[i] = t / Nd[ND_L + i * NV + i];
      }
      for (int i = 0; i < NV; ++i) {
        if (j < NA) Nd[ND_K + i * NA + j] = -y[i];
        else Nd[ND_KV + i] = -y[i];
      }
    }
    T.sync();
    // (e) value function in closed-loop ("Joseph") form.  P = Lc' Qown Lc + sum_e Acl' P_c Acl with
    //     Lc = [I;K], Acl = Atilde*Lc:  the huge Sigma entries of active state bounds inside P_c are
    //     multiplied by closed-loop maps that are ~0 in the constrained directions, instead of being
    //     cancelled against each other as in Qxx - Qxv Qvv^-1 Qvx (which floors the KKT residual at
    //     ~Sigma_max * eps).
    // (e1) per child edge: Acl (NA x NA), ccl (NA)
    for (int it = T.tid; it < ne_ * NA * (NA + 1); it += T.nt) {
      const int e = e0 + it / (NA * (NA + 1));
      const int r = it % (NA * (NA + 1));
      const int i = r / (NA + 1), j = r % (NA + 1);
      double* S_ = Q.ES(e);
      const double* Nd = Q.ND(A.edge_parent[e]);
      if (j < NA) {
        double t;
        if (i < NX) {
          t = (j < NX) ? S_[ES_AB + i * NA + j] : 0.0;
          for (int u = 0; u < NU; ++u) t += S_[ES_AB + i * NA + NX + u] * Nd[ND_K + u * NA + j];
        } else {
          t = Nd[ND_K + (i - NX) * NA + j];
        }
        S_[ES_ACL + i * NA + j] = t;
      } else {
        double t;
        if (i < NX) {
          t = S_[ES_CV + i];
          for (int u = 0; u < NU; ++u) t += S_[ES_AB + i * NA + NX + u] * Nd[ND_KV + u];
        } else {
          t = Nd[ND_KV + i - NX];
        }
        S_[ES_CCL + i] = t;
      }
    }
    T.sync();
    // (e2) per child edge: TP = P_c Acl, TV = P_c ccl + p_c
    for (int it = T.tid; it < ne_ * NA * (NA + 1); it += T.nt) {
      const int e = e0 + it / (NA * (NA + 1));
      const int r = it % (NA * (NA + 1));
      const int i = r / (NA + 1), j = r % (NA + 1);
      double* S_ = Q.ES(e);
      const double* Nc = Q.ND(A.edge_child[e]);
      if (j < NA) {
        double t = 0.0;
        for (int a = 0; a < NA; ++a) t += Nc[ND_P + i * NA + a] * S_[ES_ACL + a * NA + j];
        S_[ES_TP + i * NA + j] = t;
      } else {
        double t = Nc[ND_PV + i];
        for (int a = 0; a < NA; ++a) t += Nc[ND_P + i * NA + a] * S_[ES_CCL + a];
        S_[ES_TV + i] = t;
      }
    }
    T.sync();
    // (e3) per node: P, p
    for (int it = T.tid; it < nn * NA * (NA + 1); it += T.nt) {
      const int n = n0 + it / (NA * (NA + 1));
      double* Nd = Q.ND(n);
      const int r = it % (NA * (NA + 1));
      const int i = r / (NA + 1), j = r % (NA + 1);
      const int cs = A.node_child_start[n], cc = A.node_child_count[n];
      // column i of Lc = [e_i ; K[:,i]]
      if (j < NA) {
        double t = Nd[ND_QO + i * NYT + j];
        for (int q = 0; q < NV; ++q) {
          t += Nd[ND_QO + i * NYT + NA + q] * Nd[ND_K + q * NA + j];
          t += Nd[ND_K + q * NA + i] * Nd[ND_QO + (NA + q) * NYT + j];
          double t2 = 0.0;
          for (int w = 0; w < NV; ++w) t2 += Nd[ND_QO + (NA + q) * NYT + NA + w] * Nd[ND_K + w * NA + j];
          t += Nd[ND_K + q * NA + i] * t2;
        }
        for (int c = 0; c < cc; ++c) {
          const double* S_ = Q.ES(cs + c);
          for (int a = 0; a < NA; ++a) t += S_[ES_ACL + a * NA + i] * S_[ES_TP + a * NA + j];
        }
        Nd[ND_P + i * NA + j] = t;
      } else {
        // p = Lc' (Qown l0 + qown) + sum Acl' (P_c ccl + p_c),  l0 = [0; kv]
        double t = Nd[ND_QOV + i];
        for (int w = 0; w < NV; ++w) t += Nd[ND_QO + i * NYT + NA + w] * Nd[ND_KV + w];
        for (int q = 0; q < NV; ++q) {
          double t2 = Nd[ND_QOV + NA + q];
          for (int w = 0; w < NV; ++w) t2 += Nd[ND_QO + (NA + q) * NYT + NA + w] * Nd[ND_KV + w];
          t += Nd[ND_K + q * NA + i] * t2;
        }
        for (int c = 0; c < cc; ++c) {
          const double* S_ = Q.ES(cs + c);
          for (int a = 0; a < NA; ++a) t += S_[ES_ACL + a * NA + i] * S_[ES_TV + a];
        }
        Nd[ND_PV + i] = t;
      }
    }
    T.sync();
  }
  return 0;
}

// Forward sweep: steps for node variables, then per edge the collocation steps and multipliers.
DOMPC_DEV inline void riccati_forward(const Thr& T, const Prob& Q, double mu, double delta) {
  const KArgs& A = *Q.A;
  // root
  if (T.tid == 0) {
    double* Nd = Q.ND(0);
    const int xo = A.node_x_off[0];
    for (int a = 0; a < NX; ++a) { Nd[ND_DXT + a] = -Q.c[a]; Q.dx[xo + a] = -Q.c[a]; }
    for (int a = NX; a < NA; ++a) Nd[ND_DXT + a] = 0.0;
  }
  T.sync();
  for (int k = 0; k < A.N; ++k) {
    const int n0 = A.level_node_start[k], n1 = A.level_node_start[k + 1];
    for (int n = n0 + T.tid; n < n1; n += T.nt) {
      double* Nd = Q.ND(n);
      double dv[NV];
      for (int i = 0; i < NV; ++i) {
        double t = Nd[ND_KV + i];
        for (int a = 0; a < NA; ++a) t += Nd[ND_K + i * NA + a] * Nd[ND_DXT + a];
        dv[i] = t;
      }
      const int uo = A.node_u_off[n];
      for (int i = 0; i < NU; ++i) Q.dx[uo + i] = dv[i];
      if (NS > 0) for (int q = 0; q < NS; ++q) Q.dx[A.node_eps_off[n] + q] = dv[NU + q];
      const int cs = A.node_child_start[n], cc = A.node_child_count[n];
      for (int c = 0; c < cc; ++c) {
        const int e = cs + c, cn = A.edge_child[e];
        const double* S_ = Q.ES(e);
        double* Nc = Q.ND(cn);
        for (int a = 0; a < NX; ++a) {
          double t = S_[ES_CV + a];
          for (int b = 0; b < NX; ++b) t += S_[ES_AB + a * NA + b] * Nd[ND_DXT + b];
          for (int b = 0; b < NU; ++b) t += S_[ES_AB + a * NA + NX + b] * dv[b];
          Nc[ND_DXT + a] = t;
          Q.dx[A.node_x_off[cn] + a] = t;
        }
        for (int b = 0; b < NU; ++b) Nc[ND_DXT + NX + b] = dv[b];
      }
    }
    T.sync();
  }
  // initial-condition multiplier step
  if (T.tid == 0) {
    const double* Nd = Q.ND(0);
    for (int a = 0; a < NX; ++a) {
      double t = Nd[ND_PV + a];
      for (int b = 0; b < NA; ++b) t += Nd[ND_P + a * NA + b] * Nd[ND_DXT + b];
      Q.dlam[a] = -t;
    }
  }
  // per edge: dw, d nu, d lambda, nl_cons steps
  for (int e = T.tid; e < A.n_edges; e += T.nt) {
    const int n = A.edge_parent[e], cn = A.edge_child[e];
    const double* Nd = Q.ND(n);
    const double* Nc = Q.ND(cn);
    const int row0 = A.edge_row0[e];
    double dy[NA];
    for (int a = 0; a < NX; ++a) dy[a] = Nd[ND_DXT + a];
    for (int b = 0; b < NU; ++b) dy[NX + b] = Q.dx[A.node_u_off[n] + b];
    double dnu[NX];
    for (int a = 0; a < NX; ++a) {
      double t = Nc[ND_PV + a];
      for (int b = 0; b < NA; ++b) t += Nc[ND_P + a * NA + b] * Nc[ND_DXT + b];
      dnu[a] = t;
      Q.dlam[row0 + NW + a] = t;
    }
    if (M > 0) {
      const int woff = A.edge_w_off[e];
      double dw[NW1], rhs[NW1];
      for (int r = 0; r < NW; ++r) {
        double t = Q.EW(e, EW_W0 + r);
        for (int b = 0; b < NA; ++b) t += Q.EW(e, EW_W + r * NA + b) * dy[b];
        dw[r] = t;
        Q.dx[woff + r] = t;
      }
      // rhs = -(rw + (Hww+delta) dw + Hwu du + S' dnu)
      for (int r = 0; r < NW; ++r) {
        double t = Q.EW(e, EW_RW + r) + (Q.EW(e, EW_SIGW + r) + delta) * dw[r];
        if (r >= (M - 1) * NX) t += dnu[r - (M - 1) * NX];
        rhs[r] = t;
      }
      for (int i = 0; i < NI; ++i)
        for (int j = 1; j <= DEG; ++j) {
          const int sl = slot_of(i, j), pt = i * DEG + (j - 1);
          for (int a = 0; a < NX; ++a) {
            double t = 0.0;
            for (int b = 0; b < NX; ++b) t += Q.EW(e, EW_HP + pt * NA * NA + a * NA + b) * dw[sl * NX + b];
            for (int b = 0; b < NU; ++b) t += Q.EW(e, EW_HP + pt * NA * NA + a * NA + NX + b) * dy[NX + b];
            rhs[sl * NX + a] += t;
          }
        }
      for (int r = 0; r < NW; ++r) rhs[r] = -rhs[r];
      // solve G_w' dl = rhs with G_w = P^T L U:  U' y = rhs ; L' z = y ; dl = P^T z
      for (int r = 0; r < NW; ++r) {
        double t = rhs[r];
        for (int q = 0; q < r; ++q) t -= Q.EW(e, EW_LU + q * NW + r) * rhs[q];
        rhs[r] = t / Q.EW(e, EW_LU + r * NW + r);
      }
      for (int r = NW - 1; r >= 0; --r) {
        double t = rhs[r];
        for (int q = r + 1; q < NW; ++q) t -= Q.EW(e, EW_LU + q * NW + r) * rhs[q];
        rhs[r] = t;
      }
      for (int r = NW - 1; r >= 0; --r) {
        const int pv = (int)Q.EW(e, EW_PIV + r);
        if (pv != r) { const double t = rhs[r]; rhs[r] = rhs[pv]; rhs[pv] = t; }
      }
      for (int r = 0; r < NW; ++r) Q.dlam[row0 + r] = rhs[r];
    }
    if (NE > 0) {
      const double* S_ = Q.ES(e);
      for (int i = 0; i < NE; ++i) {
        double t = S_[ES_RDN + i];
        for (int b = 0; b < NA; ++b) t += Q.EW(e, EW_JD + i * NA + b) * dy[b];
        if (DOMPC_NL_SLACK[i] >= 0) t -= Q.dx[A.node_eps_off[n] + DOMPC_NL_SLACK[i]];
        Q.ds[e * NE1 + i] = t;
        Q.dlam[row0 + NW + NX + i] = (S_[ES_SIGS + i] + delta) * t + S_[ES_RSN + i];
      }
    }
  }
  // dummies (variables in no constraint / cost): independent scalar Newton steps
  for (int d = T.tid; d < A.n_dummy; d += T.nt) {
    const int g = A.dummy_idx[d];
    const double sg = sigma_of(Q.x[g], Q.lb[g], Q.ub[g], Q.zl[g], Q.zu[g]) + delta;
    Q.dx[g] = sg > 0.0 ? -bar_grad(Q.x[g], Q.lb[g], Q.ub[g], mu) / sg : 0.0;
  }
  T.sync();
  // bound multiplier steps
  for (int g = T.tid; g < A.n_opt_x; g += T.nt) {
    const double xv = Q.x[g], l = Q.lb[g], u = Q.ub[g];
    Q.dzl[g] = (l > -INFINITY) ? mu / (xv - l) - Q.zl[g] - Q.zl[g] / (xv - l) * Q.dx[g] : 0.0;
    Q.dzu[g] = (u < INFINITY) ? mu / (u - xv) - Q.zu[g] + Q.zu[g] / (u - xv) * Q.dx[g] : 0.0;
  }
  for (int g = T.tid; g < A.n_edges * NE; g += T.nt) {
    const int si = (g / NE1) * NE1 + g % NE1;
    const double sv = Q.s[si], l = Q.sl[si], u = Q.su[si];
    Q.dzsl[si] = (l > -INFINITY) ? mu / (sv - l) - Q.zsl[si] - Q.zsl[si] / (sv - l) * Q.ds[si] : 0.0;
    Q.dzsu[si] = (u < INFINITY) ? mu / (u - sv) - Q.zsu[si] + Q.zsu[si] / (u - sv) * Q.ds[si] : 0.0;
  }
  T.sync();
}

// ================================================================================================
struct Errs { double e_d, e_p, e_c0, sum_y, sum_z, obj, theta; };

// derivative sweep at the current iterate: per-edge evaluation/condensing, node assembly, dummies
DOMPC_DEV inline int sweep(const Thr& T, Prob& Q, double mu) {
  const KArgs& A = *Q.A;
  if (T.tid == 0) T.flags[1] = 0;
  T.sync();
  for (int g = T.tid; g < NX; g += T.nt) Q.c[g] = Q.x[A.node_x_off[0] + g] - Q.P[g] / DOMPC_SX[g];
  for (int e = T.tid; e < A.n_edges; e += T.nt)
    if (eval_edge(Q, e, mu)) T.flags[1] = 1;
  T.sync();
  for (int n = T.tid; n < A.n_nodes; n += T.nt) assemble_node(Q, n);
  for (int d = T.tid; d < A.n_dummy; d += T.nt) {
    const int g = A.dummy_idx[d];
    Q.gf[g] = 0.0;
    Q.rd[g] = -Q.zl[g] + Q.zu[g];
  }
  T.sync();
  return T.flags[1];
}

// error measures (IPOPT eq. (5)/(6)) + objective + theta at the current iterate
DOMPC_DEV inline Errs measure(const Thr& T, const Prob& Q, double mu_c) {
  const KArgs& A = *Q.A;
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // e_d, e_p, e_c, sum|y|, sum z, obj, theta
  for (int g = T.tid; g < A.n_opt_x; g += T.nt) {
    v[0] = fmax(v[0], fabs(Q.rd[g]));
    const double l = Q.lb[g], u = Q.ub[g];
    if (l > -INFINITY) { v[2] = fmax(v[2], fabs((Q.x[g] - l) * Q.zl[g] - mu_c)); v[4] += Q.zl[g]; }
    if (u < INFINITY) { v[2] = fmax(v[2], fabs((u - Q.x[g]) * Q.zu[g] - mu_c)); v[4] += Q.zu[g]; }
  }
  for (int g = T.tid; g < A.n_edges * NE; g += T.nt) {
    const int e = g / NE1, i = g % NE1;
    const int si = e * NE1 + i;
    const double yd = Q.lam[A.edge_row0[e] + NW + NX + i];
    v[0] = fmax(v[0], fabs(-yd - Q.zsl[si] + Q.zsu[si]));
    const double l = Q.sl[si], u = Q.su[si];
    if (l > -INFINITY) { v[2] = fmax(v[2], fabs((Q.s[si] - l) * Q.zsl[si] - mu_c)); v[4] += Q.zsl[si]; }
    if (u < INFINITY) { v[2] = fmax(v[2], fabs((u - Q.s[si]) * Q.zsu[si] - mu_c)); v[4] += Q.zsu[si]; }
  }
  for (int r = T.tid; r < A.n_g; r += T.nt) {
    v[1] = fmax(v[1], fabs(Q.c[r]));
    v[3] += fabs(Q.lam[r]);
    v[6] += fabs(Q.c[r]);
  }
  for (int e = T.tid; e < A.n_edges; e += T.nt) v[5] += Q.ES(e)[ES_OBJ];
  for (int n = T.tid; n < A.n_nodes; n += T.nt) v[5] += node_rterm_f(Q, n, Q.x);
  const int ops[8] = {R_MAX, R_MAX, R_MAX, R_SUM, R_SUM, R_SUM, R_SUM, R_SUM};
  wg_reduce(T, v, ops);
  Errs E;
  E.e_d = v[0]; E.e_p = v[1]; E.e_c0 = v[2]; E.sum_y = v[3]; E.sum_z = v[4]; E.obj = v[5]; E.theta = v[6];
  return E;
}

// ================================================================================================
DOMPC_DEV inline void solve_problem(const Thr& T, const KArgs& A, int b, int slot) {
  const dompc_options& O = A.opt;
  Prob Q = make_prob(A, slot, A.p + (int64_t)b * A.n_opt_p);
  const double* x0 = A.x0 + (int64_t)b * A.n_opt_x;
  const int nX = A.n_opt_x, nSl = A.n_edges * NE;
  int status = 2, it = 0, n_reg = 0, n_ls_fail = 0, n_sweeps = 0, n_trials = 0;

  // ---- bounds (relaxed, bound_relax_factor), starting point pushed inside, z = 1
  double cnt[2] = {0.0, 0.0};
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
    Q.lb[g] = l; Q.ub[g] = u; Q.x[g] = xv;
    Q.zl[g] = hl ? 1.0 : 0.0; Q.zu[g] = hu ? 1.0 : 0.0;
    cnt[0] += (hl ? 1.0 : 0.0) + (hu ? 1.0 : 0.0);
  }
  for (int r = T.tid; r < A.n_g; r += T.nt) Q.lam[r] = 0.0;
  T.sync();
  // slacks of the nl_cons rows: s = d(x) pushed into [lbg,ubg]
  if (NE > 0) {
    for (int e = T.tid; e < A.n_edges; e += T.nt) {
      for (int i = 0; i < NE; ++i) Q.s[e * NE1 + i] = 0.0;
      eval_edge_f(Q, e, Q.x, Q.s, Q.ct);
      for (int i = 0; i < NE; ++i) {
        const int row = A.edge_row0[e] + NW + NX + i, si = e * NE1 + i;
        double l = A.lbg[row], u = A.ubg[row];
        if (l > -INFINITY) l -= fmin(O.constr_viol_tol, O.bound_relax_factor * fmax(1.0, fabs(l)));
        if (u < INFINITY) u += fmin(O.constr_viol_tol, O.bound_relax_factor * fmax(1.0, fabs(u)));
        const bool hl = l > -INFINITY, hu = u < INFINITY;
        double pl = hl ? O.bound_push * fmax(1.0, fabs(l)) : 0.0;
        double pu = hu ? O.bound_push * fmax(1.0, fabs(u)) : 0.0;
        if (hl && hu) { pl = fmin(pl, O.bound_frac * (u - l)); pu = fmin(pu, O.bound_frac * (u - l)); }
        double sv = Q.ct[row];       // = d - 0
        if (hl) sv = fmax(sv, l + pl);
        if (hu) sv = fmin(sv, u - pu);
        Q.s[si] = sv; Q.sl[si] = l; Q.su[si] = u;
        Q.zsl[si] = hl ? 1.0 : 0.0; Q.zsu[si] = hu ? 1.0 : 0.0;
        cnt[1] += (hl ? 1.0 : 0.0) + (hu ? 1.0 : 0.0);
      }
    }
    T.sync();
  }
  {
    const int ops[2] = {R_SUM, R_SUM};
    wg_reduce(T, cnt, ops);
  }
  const double n_bounds = cnt[0] + cnt[1];
  const double n_dual = (double)A.n_g + n_bounds;

  // ---- objective scaling from the gradient at the (pushed) starting point
  double mu = O.mu_init;
  Q.sf = 1.0;
  int bad = sweep(T, Q, mu);
  ++n_sweeps;
  if (O.obj_scaling) {
    double gm[1] = {0.0};
    for (int g = T.tid; g < nX; g += T.nt) gm[0] = fmax(gm[0], fabs(Q.gf[g]));
    const int ops[1] = {R_MAX};
    wg_reduce(T, gm, ops);
    if (gm[0] > O.nlp_scaling_max_gradient) {
      Q.sf = fmax(O.nlp_scaling_max_gradient / gm[0], 1e-8);
      bad = sweep(T, Q, mu);
      ++n_sweeps;
    }
  }
  const double mu_min = fmin(O.tol, O.compl_inf_tol * Q.sf) / (O.kappa_eps + 1.0);
  double tau = fmax(O.tau_min, 1.0 - mu);
  Errs E = measure(T, Q, 0.0);
  const double theta0 = E.theta;
  const double theta_max = 1e4 * fmax(1.0, theta0), theta_min = 1e-4 * fmax(1.0, theta0);
  int n_filt = 0;
  double delta_last = 0.0;
  int acc_count = 0;
  const double s_max = 100.0;
  double E0 = 0.0;

  while (true) {
    if (bad) { status = 3; break; }
    const double sd = fmax(s_max, (E.sum_y + E.sum_z) / fmax(1.0, n_dual)) / s_max;
    const double sc = fmax(s_max, E.sum_z / fmax(1.0, n_bounds)) / s_max;
    E0 = fmax(E.e_d / sd, fmax(E.e_p, E.e_c0 / sc));
    if (!(E0 == E0) || !(E.obj == E.obj)) { status = 4; break; }
    if (E0 <= O.tol && E.e_d <= O.dual_inf_tol && E.e_p <= O.constr_viol_tol && E.e_c0 <= O.compl_inf_tol) {
      status = 0; break;
    }
    if (E0 <= O.acceptable_tol) {
      if (++acc_count >= O.acceptable_iter) { status = 1; break; }
    } else acc_count = 0;
    if (it >= O.max_iter) { status = 2; break; }

    // ---- barrier update (monotone Fiacco-McCormick)
    bool mu_changed = false;
    while (true) {
      Errs Em = measure(T, Q, mu);
      const double Emu = fmax(Em.e_d / sd, fmax(Em.e_p, Em.e_c0 / sc));
      if (Emu <= O.kappa_eps * mu && mu > mu_min) {
        mu = fmax(mu_min, fmin(O.kappa_mu * mu, pow(mu, O.theta_mu)));
        tau = fmax(O.tau_min, 1.0 - mu);
        n_filt = 0;
        mu_changed = true;
      } else break;
    }
    if (mu_changed) { bad = sweep(T, Q, mu); ++n_sweeps; if (bad) { status = 3; break; } }

    // ---- search direction with inertia correction (delta_w on all primal variables)
    double delta = 0.0;
    bool first_try = true, dir_ok = true;
    while (true) {
      const int fail = riccati_backward(T, Q, mu, delta);
      if (!fail) break;
      if (delta == 0.0) {
        delta = (delta_last == 0.0) ? O.delta_w_0 : fmax(O.delta_w_min, O.kappa_w_minus * delta_last);
      } else {
        delta *= (delta_last == 0.0 && first_try) ? O.kappa_w_plus_bar : O.kappa_w_plus;
        first_try = false;   // (IPOPT: the larger factor only on the very first increase)
        if (delta > O.delta_w_max) { dir_ok = false; break; }
      }
    }
    if (!dir_ok) { status = 3; break; }
    if (delta > 0.0) { delta_last = delta; ++n_reg; }
    riccati_forward(T, Q, mu, delta);

    // ---- fraction to the boundary, directional derivative of the barrier function
    double r5[5] = {1.0, 1.0, 0.0, 0.0, 0.0};   // a_max, a_z, dphi, barrier-sum, (unused)
    for (int g = T.tid; g < nX; g += T.nt) {
      const double xv = Q.x[g], l = Q.lb[g], u = Q.ub[g], d = Q.dx[g];
      double gphi = Q.gf[g];
      if (l > -INFINITY) {
        if (d < 0.0) r5[0] = fmin(r5[0], -tau * (xv - l) / d);
        if (Q.dzl[g] < 0.0) r5[1] = fmin(r5[1], -tau * Q.zl[g] / Q.dzl[g]);
        gphi -= mu / (xv - l);
        r5[3] -= log(xv - l);
      }
      if (u < INFINITY) {
        if (d > 0.0) r5[0] = fmin(r5[0], tau * (u - xv) / d);
        if (Q.dzu[g] < 0.0) r5[1] = fmin(r5[1], -tau * Q.zu[g] / Q.dzu[g]);
        gphi += mu / (u - xv);
        r5[3] -= log(u - xv);
      }
      r5[2] += gphi * d;
    }
    for (int g = T.tid; g < nSl; g += T.nt) {
      const int si = (g / NE1) * NE1 + g % NE1;
      const double sv = Q.s[si], l = Q.sl[si], u = Q.su[si], d = Q.ds[si];
      double gphi = 0.0;
      if (l > -INFINITY) {
        if (d < 0.0) r5[0] = fmin(r5[0], -tau * (sv - l) / d);
        if (Q.dzsl[si] < 0.0) r5[1] = fmin(r5[1], -tau * Q.zsl[si] / Q.dzsl[si]);
        gphi -= mu / (sv - l);
        r5[3] -= log(sv - l);
      }
      if (u < INFINITY) {
        if (d > 0.0) r5[0] = fmin(r5[0], tau * (u - sv) / d);
        if (Q.dzsu[si] < 0.0) r5[1] = fmin(r5[1], -tau * Q.zsu[si] / Q.dzsu[si]);
        gphi += mu / (u - sv);
        r5[3] -= log(u - sv);
      }
      r5[2] += gphi * d;
    }
    {
      const int ops[5] = {R_MIN, R_MIN, R_SUM, R_SUM, R_SUM};
      wg_reduce(T, r5, ops);
    }
    const double a_max = r5[0], a_z = r5[1], dphi = r5[2];
    const double theta = E.theta;
    const double phi = E.obj + mu * r5[3];

    // ---- filter line search (no second-order correction, no restoration phase)
    const double gamma_theta = 1e-5, gamma_phi = 1e-8, eta_phi = 1e-8, s_theta = 1.1, s_phi = 2.3, gamma_alpha = 0.05;
    double a_min;
    if (dphi < 0.0 && theta <= theta_min)
      a_min = (theta > 0.0) ? gamma_alpha * fmin(gamma_theta, fmin(gamma_phi * theta / (-dphi),
                                                                    pow(theta, s_theta) / pow(-dphi, s_phi)))
                            : gamma_alpha * gamma_theta;
    else if (dphi < 0.0) a_min = gamma_alpha * fmin(gamma_theta, gamma_phi * theta / (-dphi));
    else a_min = gamma_alpha * gamma_theta;
    a_min = fmax(a_min, 1e-14);
    double alpha = a_max;
    bool accepted = false, armijo_used = false;
    double th_t = 0.0, obj_t = 0.0;
    while (true) {
      for (int g = T.tid; g < nX; g += T.nt) Q.xt[g] = Q.x[g] + alpha * Q.dx[g];
      for (int g = T.tid; g < nSl; g += T.nt) {
        const int si = (g / NE1) * NE1 + g % NE1;
        Q.st[si] = Q.s[si] + alpha * Q.ds[si];
      }
      T.sync();
      double r3[3] = {0.0, 0.0, 0.0};    // obj, theta, barrier
      for (int g = T.tid; g < NX; g += T.nt) Q.ct[g] = Q.xt[A.node_x_off[0] + g] - Q.P[g] / DOMPC_SX[g];
      for (int e = T.tid; e < A.n_edges; e += T.nt) r3[0] += eval_edge_f(Q, e, Q.xt, Q.st, Q.ct);
      for (int n = T.tid; n < A.n_nodes; n += T.nt) r3[0] += node_rterm_f(Q, n, Q.xt);
      T.sync();
      for (int r = T.tid; r < A.n_g; r += T.nt) r3[1] += fabs(Q.ct[r]);
      for (int g = T.tid; g < nX; g += T.nt) {
        const double l = Q.lb[g], u = Q.ub[g];
        if (l > -INFINITY) r3[2] -= log(Q.xt[g] - l);
        if (u < INFINITY) r3[2] -= log(u - Q.xt[g]);
      }
      for (int g = T.tid; g < nSl; g += T.nt) {
        const int si = (g / NE1) * NE1 + g % NE1;
        if (Q.sl[si] > -INFINITY) r3[2] -= log(Q.st[si] - Q.sl[si]);
        if (Q.su[si] < INFINITY) r3[2] -= log(Q.su[si] - Q.st[si]);
      }
      {
        const int ops[3] = {R_SUM, R_SUM, R_SUM};
        wg_reduce(T, r3, ops);
      }
      ++n_trials;
      obj_t = r3[0]; th_t = r3[1];
      const double ph_t = obj_t + mu * r3[2];
      bool ok = (ph_t == ph_t) && (th_t == th_t) && fabs(ph_t) < INFINITY && th_t <= theta_max;
      if (ok) {
        for (int q = 0; q < n_filt; ++q)
          if (th_t >= T.filt[2 * q] && ph_t >= T.filt[2 * q + 1]) { ok = false; break; }
      }
      bool armijo_case = false;
      if (ok) {
        const bool switching = dphi < 0.0 && alpha * pow(-dphi, s_phi) > pow(theta, s_theta);
        const double eps_m = 10.0 * 2.220446049250313e-16 * fabs(phi);
        if (theta <= theta_min && switching) {
          armijo_case = true;
          ok = (ph_t - phi - eps_m <= eta_phi * alpha * dphi);
        } else {
          ok = (th_t <= (1.0 - gamma_theta) * theta) || (ph_t - phi - eps_m <= -gamma_phi * theta);
        }
      }
      if (ok) { accepted = true; armijo_used = armijo_case; break; }
      if (alpha * 0.5 < a_min) break;      // xt/st/ct stay at the last evaluated alpha
      alpha *= 0.5;
    }
    if (!accepted) {
      // no restoration phase: take the smallest trial step and reset the filter
      ++n_ls_fail;
      n_filt = 0;
    } else if (!armijo_used) {
      if (T.tid == 0) {
        int q = n_filt < MAX_FILTER ? n_filt : MAX_FILTER - 1;
        T.filt[2 * q] = (1.0 - gamma_theta) * theta;
        T.filt[2 * q + 1] = phi - gamma_phi * theta;
      }
      if (n_filt < MAX_FILTER) ++n_filt;
      T.sync();
    }
    // ---- accept the trial point
    const double ks = 1e10;
    for (int g = T.tid; g < nX; g += T.nt) {
      const double xv = Q.xt[g];
      Q.x[g] = xv;
      const double l = Q.lb[g], u = Q.ub[g];
      if (l > -INFINITY) {
        double z = Q.zl[g] + a_z * Q.dzl[g];
        const double d = xv - l;
        Q.zl[g] = fmax(fmin(z, ks * mu / d), mu / (ks * d));
      }
      if (u < INFINITY) {
        double z = Q.zu[g] + a_z * Q.dzu[g];
        const double d = u - xv;
        Q.zu[g] = fmax(fmin(z, ks * mu / d), mu / (ks * d));
      }
    }
    for (int g = T.tid; g < nSl; g += T.nt) {
      const int si = (g / NE1) * NE1 + g % NE1;
      const double sv = Q.st[si];
      Q.s[si] = sv;
      const double l = Q.sl[si], u = Q.su[si];
      if (l > -INFINITY) {
        double z = Q.zsl[si] + a_z * Q.dzsl[si];
        Q.zsl[si] = fmax(fmin(z, ks * mu / (sv - l)), mu / (ks * (sv - l)));
      }
      if (u < INFINITY) {
        double z = Q.zsu[si] + a_z * Q.dzsu[si];
        Q.zsu[si] = fmax(fmin(z, ks * mu / (u - sv)), mu / (ks * (u - sv)));
      }
    }
    for (int r = T.tid; r < A.n_g; r += T.nt) Q.lam[r] += alpha * Q.dlam[r];
    if (A.trace && b == 0 && T.tid == 0 && it < A.trace_cap) {
      double* tr = A.trace + 8 * it;
      tr[0] = it; tr[1] = mu; tr[2] = E0; tr[3] = E.e_p; tr[4] = E.e_d; tr[5] = accepted ? alpha : -alpha;
      tr[6] = delta; tr[7] = E.obj / Q.sf;
    }
    T.sync();
    ++it;
    bad = sweep(T, Q, mu);
    ++n_sweeps;
    E = measure(T, Q, 0.0);
  }

  // ---- outputs (unscaled multipliers, CasADi sign convention)
  const double isf = 1.0 / Q.sf;
  if (A.x_out) for (int g = T.tid; g < nX; g += T.nt) A.x_out[(int64_t)b * nX + g] = Q.x[g];
  if (A.lam_x_out) for (int g = T.tid; g < nX; g += T.nt) A.lam_x_out[(int64_t)b * nX + g] = (Q.zu[g] - Q.zl[g]) * isf;
  if (A.lam_g_out) for (int r = T.tid; r < A.n_g; r += T.nt) A.lam_g_out[(int64_t)b * A.n_g + r] = Q.lam[r] * isf;
  if (A.g_out) {
    // g in the reference's convention: equality rows = residual (+rhs 0), nl rows = d(x)
    for (int r = T.tid; r < A.n_g; r += T.nt) A.g_out[(int64_t)b * A.n_g + r] = Q.c[r];
    T.sync();
    for (int g = T.tid; g < nSl; g += T.nt) {
      const int e = g / NE1, i = g % NE1;
      const int row = A.edge_row0[e] + NW + NX + i;
      A.g_out[(int64_t)b * A.n_g + row] = Q.c[row] + Q.s[e * NE1 + i];
    }
  }
  if (T.tid == 0) {
    if (A.f_out) A.f_out[b] = E.obj * isf;
    if (A.stats) {
      dompc_stats& S = A.stats[b];
      S.success = (status == 0 || status == 1) ? 1 : 0;
      S.status = status; S.iter_count = it; S.n_reg = n_reg; S.n_ls_fail = n_ls_fail; S.n_sweeps = n_sweeps; S.n_trials = n_trials; S.reserved = 0;
      S.mu = mu; S.obj = E.obj * isf; S.inf_pr = E.e_p; S.inf_du = E.e_d; S.inf_compl = E.e_c0;
      S.obj_scaling = Q.sf; S.t_wall_total = 0.0;
    }
  }
  T.sync();
}

// ------------------------------------------------------------------------------------------------
// mode 1: one Newton direction at a given primal-dual point (parity tests against the oracle's
// sparse KKT solve).  Slacks: s = d(x) pushed inside, z_s = 1.
DOMPC_DEV inline void debug_newton(const Thr& T, const KArgs& A) {
  const dompc_options& O = A.opt;
  Prob Q = make_prob(A, 0, A.p);
  const int nX = A.n_opt_x;
  for (int g = T.tid; g < nX; g += T.nt) {
    Q.x[g] = A.x0[g]; Q.lb[g] = A.lbx[g]; Q.ub[g] = A.ubx[g];
    Q.zl[g] = A.dbg_zl[g]; Q.zu[g] = A.dbg_zu[g];
  }
  for (int r = T.tid; r < A.n_g; r += T.nt) Q.lam[r] = A.dbg_lam[r];
  T.sync();
  if (NE > 0) {
    for (int e = T.tid; e < A.n_edges; e += T.nt) {
      for (int i = 0; i < NE; ++i) Q.s[e * NE1 + i] = 0.0;
      eval_edge_f(Q, e, Q.x, Q.s, Q.ct);
      for (int i = 0; i < NE; ++i) {
        const int row = A.edge_row0[e] + NW + NX + i, si = e * NE1 + i;
        const double l = A.lbg[row], u = A.ubg[row];
        const bool hl = l > -INFINITY, hu = u < INFINITY;
        double pl = hl ? O.bound_push * fmax(1.0, fabs(l)) : 0.0;
        double pu = hu ? O.bound_push * fmax(1.0, fabs(u)) : 0.0;
        if (hl && hu) { pl = fmin(pl, O.bound_frac * (u - l)); pu = fmin(pu, O.bound_frac * (u - l)); }
        double sv = Q.ct[row];
        if (hl) sv = fmax(sv, l + pl);
        if (hu) sv = fmin(sv, u - pu);
        Q.s[si] = sv; Q.sl[si] = l; Q.su[si] = u;
        Q.zsl[si] = hl ? 1.0 : 0.0; Q.zsu[si] = hu ? 1.0 : 0.0;
      }
    }
    T.sync();
  }
  Q.sf = 1.0;
  sweep(T, Q, A.dbg_mu);
  const int fail = riccati_backward(T, Q, A.dbg_mu, A.dbg_delta);
  riccati_forward(T, Q, A.dbg_mu, A.dbg_delta);
  for (int g = T.tid; g < nX; g += T.nt) {
    A.dbg_dx[g] = fail ? NAN : Q.dx[g];
    A.dbg_rd[g] = Q.rd[g];
  }
  for (int r = T.tid; r < A.n_g; r += T.nt) { A.dbg_dlam[r] = Q.dlam[r]; A.dbg_c[r] = Q.c[r]; }
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
    Q.s[si] = 0.0; Q.sl[si] = -INFINITY; Q.su[si] = INFINITY; Q.zsl[si] = 0.0; Q.zsu[si] = 0.0;
  }
  T.sync();
  Q.sf = 1.0;
  sweep(T, Q, 0.0);
  double* gout = A.sw_g + (int64_t)b * A.n_g;
  for (int r = T.tid; r < A.n_g; r += T.nt) gout[r] = Q.c[r];
  double* bl = A.sw_blocks + (int64_t)b * A.n_edges * SWEEP_BLOCK;
  for (int it = T.tid; it < A.n_edges * SWEEP_BLOCK; it += T.nt) {
    const int e = it / SWEEP_BLOCK, i = it % SWEEP_BLOCK;
    const double* S_ = Q.ES(e);
    double v;
    if (i < NX * NA) v = S_[ES_AB + i];
    else if (i < NX * NA + NX) v = S_[ES_CV + i - NX * NA];
    else if (i < NX * NA + NX + NA * NA) v = S_[ES_QT + i - NX * NA - NX];
    else v = S_[ES_QV + i - NX * NA - NX - NA * NA] + S_[ES_RY + i - NX * NA - NX - NA * NA];
    bl[it] = v;
  }
  T.sync();
}

}  // namespace dompc
