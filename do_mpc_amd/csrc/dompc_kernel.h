// dompc_kernel.h - structured interior-point solver for the multi-stage (scenario tree) NLP of
// do-mpc's MPC.make_step(), written for gfx950 (MI355X).  One workgroup solves one problem from
// x0-in to u0-out: model evaluation + per-edge collocation condensing, tree Riccati recursion,
// fraction-to-boundary, filter line search and barrier update all stay on the device.
//
// What it replaces in the reference (everything CasADi/IPOPT/MUMPS do inside
// `r = self.S(**kwargs)`, /root/reference/do_mpc/optimizer.py:770):
//   nlp_f / nlp_g / nlp_grad_f / nlp_jac_g / nlp_hess_l   -> eval_models() + eval_edge_coop()  (per edge, per collocation point)
//   MUMPS LDL^T of the sparse KKT matrix                  -> condense (LU of the collocation block) +
//                                                            riccati_backward()/forward() on the tree
//   IPOPT's filter line search / mu update / termination  -> solve_problem()
// The algorithm constants are IPOPT's defaults (Waechter & Biegler 2006), see include/dompc_ipm.h.  Restated details
// that decide whether the iterates (not only the limit point) are IPOPT's: the delta_w sequence on the systems that are
// singular at delta_w = 0 (`singular0`), the unused variables' barrier terms, the second-order correction, the
// least-squares multiplier estimate of the starting point - all in solve_problem(); DESIGN.md section 2.
//
// The file is compiled twice from the same text:
//   * by hipcc --offload-arch=gfx950 into the per-model code object (product path), and
//   * by g++ with -DDOMPC_HOST_EMU into a test-only library where a "workgroup" is one host
//     thread (tests/ build it; the product never loads it).
// It must be included after the generated model header (do_mpc_amd/lowering.py).
//
// Layout of the sources (round 5: this header used to hold all 6 000 lines).  This file: sizes and switches of the model, record
// layouts in HBM, the thread context `Thr` with its barriers and cross-rank exchange, reductions, the small dense products - then, included
// at its end inside namespace dompc:
//   dompc_dae.h       edge phases of models with algebraic states / rows on the edge unknowns (dense path)
//   dompc_edge.h      trial evaluation of an edge; derivative evaluation + condensing (generic path); dense image of the compact record
//   dompc_factor.h    blocked Gauss-Jordan on the FP64 matrix cores; factorisation of a single-finite-element edge; per-edge part of the sweep
//   dompc_quad.h      the edge sweep with four edges per wavefront (round 6; single finite element, no nl_cons rows)
//   dompc_node.h      gradient / dual-residual assembly of a node's variables
//   dompc_riccati.h   tree Riccati recursion, backward pass (+ dompc_riccati16.h: register-resident matrix-core recursion)
//   dompc_forward.h   forward pass, adjoint recovery of the continuity multipliers
//   dompc_sweep.h     the derivative sweep
//   dompc_phases.h    vector passes, outlined phases of the device build
//   dompc_driver.h    shared slacks, the interior-point driver solve_problem(), kernel bodies
#pragma once
#include <math.h>
#include <stdint.h>
#ifndef DOMPC_HOST_EMU
#include <utility>
#endif
#include "dompc_kargs.h"

namespace dompc {

// Building blocks of the phases.  They are `inline`; the phases themselves (phase_sweep, phase_edge_factor, phase_backward,
// phase_forward, the line-search passes) are separate NOINLINE device functions that rebuild their context (kernel
// arguments, Thr, Prob) from uniform sources - passing the descriptors by reference puts them into scratch and doubled
// the time of the LDS loops in a round-1 attempt; sharing one register allocation between the phases cost even more
// (DESIGN.md section 4).
#ifndef DOMPC_HOST_EMU
#define DOMPC_PHASE __device__ inline
#else
#define DOMPC_PHASE static inline
#endif

constexpr int NX = DOMPC_NX, NU = DOMPC_NU, NP = DOMPC_NP, NTVP = DOMPC_NTVP;
// nl_cons_single_slack (_mpc.py:1120-1123, 1228: ONE `_eps` entry per scenario slot for all stages, `eps[min(k, n_eps - 1), s]`): the slack
// variables are then shared by nodes of several stages (and, with n_robust >= 2, of several sub-trees) - no longer decision variables of a
// node.  Generated header: DOMPC_EPS_GLOBAL 1.  They leave the structured part (NS = 0: a node decides on u only; the rows and the cost
// terms still READ them, NSE) and are solved for by a Schur complement on top of the structured solve, solve_problem: eps_schur_*.
#ifndef DOMPC_EPS_GLOBAL
#define DOMPC_EPS_GLOBAL 0
#endif
constexpr bool EPS_GLOBAL = DOMPC_EPS_GLOBAL != 0 && DOMPC_NS > 0 && DOMPC_NE > 0;
constexpr int NS = EPS_GLOBAL ? 0 : DOMPC_NS;       // slack entries that are decision variables of a node
constexpr int NSE = DOMPC_NS;                       // slack entries an edge's rows / cost terms read (at node_eps_off of its parent node)
constexpr int NVG_MAX = 32;                         // EPS_GLOBAL: at most this many shared slack variables (n_eps * S * ns)
constexpr int NZ = DOMPC_NZ;
constexpr int DEG = DOMPC_DEG, NI = DOMPC_NI, M = DOMPC_M;
// nl_cons rows of an edge: ONE evaluation of the user's expressions at (x_n, u, z of the first point) (_mpc.py:1239-1246) or,
// with nl_cons_check_colloc_points, one evaluation per stored point i of the interval at (_x[k+1,s,i], u, _z[k,s,i])
// (_mpc.py:1229-1237; chain problems only: the reference indexes the points with the PARENT's scenario).  Rows that depend on
// the edge unknowns are handled by the dense edge path (dompc_dae.h), like the algebraic states.
#ifndef DOMPC_NL_COLLOC
#define DOMPC_NL_COLLOC 0
#endif
// Estimators (moving horizon estimation, _mhe.py:1030-1211) run on the same kernels with three switches of the generated header:
//   DOMPC_FREE_ROOT  the initial state is a free variable with an arrival cost dompc_aterm(x_0; previous estimate in the `_x0` slot of
//                    opt_p) instead of the initial-condition rows (which stay in g as 0 = 0, multipliers 0);
//   DOMPC_LT_END     the stage cost of an edge reads the END state of the interval (the measurement residual of stage k);
//   DOMPC_NL_DUP     the nl_cons rows of the last evaluated point appear a second time (_mhe.py:1186-1188).
#ifndef DOMPC_FREE_ROOT
#define DOMPC_FREE_ROOT 0
#endif
#ifndef DOMPC_LT_END
#define DOMPC_LT_END 0
#endif
#ifndef DOMPC_NL_DUP
#define DOMPC_NL_DUP 0
#endif
#if !DOMPC_FREE_ROOT
DOMPC_FN double dompc_aterm_f(const double*, const double*, const double*, const double*) { return 0.0; }
DOMPC_FN void dompc_aterm(const double*, const double*, const double*, const double*, double*, double*, double*) {}
#endif
constexpr bool FREE_ROOT = DOMPC_FREE_ROOT != 0, LT_END = DOMPC_LT_END != 0 && M > 0;
constexpr int NEB = DOMPC_NE;                               // rows of one evaluation
constexpr int NLP = (DOMPC_NL_COLLOC && M > 0) ? M : 1;     // points at which the rows are evaluated
constexpr bool NL_DUP = DOMPC_NL_DUP != 0 && NEB > 0;
constexpr int NLB = NLP + (NL_DUP ? 1 : 0);                 // evaluations per edge
constexpr int NE = NEB * NLB;
constexpr bool NL_COLLOC = DOMPC_NL_COLLOC && M > 0 && NEB > 0;
DOMPC_HD constexpr int nl_pt(int blk) { return blk < NLP ? blk : NLP - 1; }      // point of evaluation `blk`
#ifndef DOMPC_FORCE_DENSE
#define DOMPC_FORCE_DENSE 0        // test aid: 1 = a model without algebraic states through the dense edge path as well
#endif
constexpr bool DENSE_EDGE = DOMPC_NZ > 0 || NL_COLLOC || NL_DUP || LT_END || DOMPC_FORCE_DENSE;      // edge path: dense (dompc_dae.h) instead of the single-element fast path
DOMPC_HD inline int nl_slack(int i) { return DOMPC_NL_SLACK[NLB == 1 ? i : i % NEB]; }
constexpr int NA = NX + NU;          // (x,u) of a stage == augmented state (x,u_prev)
constexpr int NV = NU + NS;          // decision variables of a node: u then eps
constexpr int NYT = NA + NV;         // node quadratic: (x, u_prev, u, eps)
// Algebraic states (DAE models, optimizer.py:905-983): every stored point of an interval has its own z (max(M, 1) per edge,
// _mpc.py:1130) and its own algebraic rows; they are edge unknowns like the collocation states and are eliminated with them
// (dense path eval_edge_dae - the optimised single-element path is for NZ == 0).
constexpr int MZ = NZ > 0 ? (M > 0 ? M : 1) : 0;     // z slots of an edge
constexpr int NWX = M * NX;          // collocation states of an edge (incl. the xkf slot)
constexpr int NW = NWX + MZ * NZ;    // eliminated unknowns of an edge = rows of its square constraint block; order: x slots, then z slots
// user-defined input penalty rterm(x, u, u_prev, tvp, p) (_mpc.py:593-677, 1263-1269) instead of the quadratic default:
// value / gradient / Hessian over r = (x, u, u_prev) per edge (lowering.py: dompc_rterm)
constexpr bool RT_CUSTOM = DOMPC_RTERM_CUSTOM != 0;
constexpr int NR = NA + NU, NR_T = NR * (NR + 1) / 2;
constexpr int RT_LEN = RT_CUSTOM ? 1 + NR + NR_T : 0;
constexpr int NF = NX + NZ;          // outputs of the dynamics at a point: [h f / sx ; alg]
constexpr int NAV = NA + NZ;         // inputs of a point function: (x, u, z)
constexpr int NCOLL = NI * DEG;      // collocation points evaluated per edge
constexpr int RPE = NW + NX + NE;    // constraint rows per edge
constexpr int NE1 = NE > 0 ? NE : 1, NEB1 = NEB > 0 ? NEB : 1;
constexpr int NS1 = NS > 0 ? NS : 1;
constexpr int NW1 = NW > 0 ? NW : 1;
constexpr int MAX_FILTER = 48;
#ifndef DOMPC_SHARD
#define DOMPC_SHARD 0
#endif
#ifndef DOMPC_GJ_U
#define DOMPC_GJ_U 0.01              // threshold of the pivot test of the collocation-block elimination in its natural order (|a_kk| >= u max|a_ik|); a huge
#endif                               // value sends every edge through the elimination with partial pivoting (test of that fallback)
#ifndef DOMPC_KO
#define DOMPC_KO 0                  // measurement aid (tools/gpu_sweep_ko.py; WRONG RESULTS): pieces of the edge sweep left out, to see what each one
#endif                              // costs in THROUGHPUT under real concurrency: 1 factorisation, 2 condensing, 4 record stores, 8 model evaluation,
                                    // 16 per-variable loads of the edge, 32 staging + expansion of the model-output record
constexpr int RED_MAX = 12;          // values reduced per pass
#ifndef DOMPC_HOST_EMU
constexpr int GS_C = 64;             // lanes per edge group: one wavefront
#else
constexpr int GS_C = 1;
#endif

// per-edge forward-pass record (contiguous per edge; written/read cooperatively by one wavefront) --
// stored part of G_w^-1 (row-major LU_N x LU_N).  Single finite element: G_w^-1 = [[Gi, 0], [-E Gi, I]] with the
// continuity rows E = -[D_1 I ... D_DEG I], so only Gi = G_cc^-1 (collocation block) is kept - 400 instead of
// 900 doubles per industrial_poly edge written by every sweep and read by every forward pass.
// Round 3: W = -G_w^-1 G_y and w0 = -G_w^-1 r are NOT stored any more (420 doubles per industrial_poly edge, written by
// every sweep and read back by every forward pass): the forward pass forms dw = -G_w^-1 (G_y dy + r) from the stored
// inverse, the u-columns of the point Jacobians (model-output record) and the residual vector.
constexpr int LU_N = (NI == 1 && DEG > 0 && !DENSE_EDGE) ? DEG * NX : NW;
constexpr int EW_LU = 0;
constexpr int EW_SIGW = EW_LU + LU_N * LU_N; // Sigma_w + dsw      (the lambda-weighted Hessian blocks are read from the model-output record)
constexpr int EW_RW = EW_SIGW + NW;
constexpr int EW_JD = EW_RW + NW;            // NE x NA
// DAE models (dense path): W, w0, the rows of the edge Hessian that belong to the eliminated unknowns (over [w | y], Sigma_w
// and the inertia correction on the diagonal), the Jacobians of the end-point rows and of the nl_cons rows w.r.t. w
constexpr int EW_W = EW_JD + NE * NA;        // NW x NA
constexpr int EW_W0 = EW_W + (DENSE_EDGE ? NW * NA : 0);
constexpr int EW_HW = EW_W0 + (DENSE_EDGE ? NW : 0);                 // NW x (NW + NA)
constexpr int EW_EWJ = EW_HW + (DENSE_EDGE ? NW * (NW + NA) : 0);    // NX x NW
constexpr int EW_JDW = EW_EWJ + (DENSE_EDGE ? NX * NW : 0);          // NE x NW
constexpr int EW_SIZE = ((EW_JDW + (DENSE_EDGE ? NE * NW : 0) + 1 + 7) / 8) * 8;      // records start on 64-byte boundaries

// per-edge shared (contiguous per edge) --------------------------------------------------------
// The head [A B | c | Q~ | q~ + r_y] is what the backward Riccati pass reads (staged by LDS-DMA, dompc_riccati16.h).
constexpr int ES_AB = 0;                     // NX x NA
constexpr int ES_CV = ES_AB + NX * NA;
constexpr int ES_QT = ES_CV + NX;            // NA x NA symmetric, packed upper triangle (symi)
constexpr int ES_QV = ES_QT + NA * (NA + 1) / 2;       // condensed gradient q~ PLUS r_y (the Riccati pass only needs the sum)       (NA)
constexpr int ES_RY = ES_QV + NA;            // G_y' lam + sf*omega*grad l + Jd' yd      (NA)   (dual residual assembly)
constexpr int ES_GFY = ES_RY + NA;           // sf*omega*grad l                            (NA)
constexpr int ES_QVB = ES_GFY + NA;          // W' b, b = barrier gradient of w per unit mu: q~(mu + dmu) = q~ + dmu W'b   (NA)
constexpr int ES_MG = ES_QVB + NA;           // sf*omega*grad m (last edges)               (NX)
constexpr int ES_MH = ES_MG + NX;            // sf*omega*hess m                            (NX x NX)
constexpr int ES_SIGS = ES_MH + NX * NX;     // NE
constexpr int ES_RDN = ES_SIGS + NE;         // d - s
constexpr int ES_RSN = ES_RDN + NE;          // -yd - mu/(s-sl) + mu/(su-s)
constexpr int ES_OBJ = ES_RSN + NE;
constexpr int ES_RTUP = ES_OBJ + 1;                          // user-defined rterm: sf*omega * d rterm / d u_prev          (NU)
constexpr int ES_RTH = ES_RTUP + (RT_CUSTOM ? NU : 0);       // ... and its Hessian over (x, u, u_prev), packed             (NR_T)
constexpr int ES_SIZE = ((ES_RTH + (RT_CUSTOM ? NR_T : 0) + 7) / 8) * 8;

// per-edge model-output record (global): results of the lowered model functions at the current iterate,
// written by the thread-parallel evaluation phase and copied into LDS by the edge groups
// symmetric blocks of the model-output record are packed (upper triangle, row by row) by the generated code
constexpr int NA_T = NA * (NA + 1) / 2, NX_T = NX * (NX + 1) / 2, NAV_T = NAV * (NAV + 1) / 2;
DOMPC_HD constexpr int symi(int i, int j, int n) { return i <= j ? i * n - i * (i - 1) / 2 + j - i : j * n - j * (j - 1) / 2 + i - j; }
constexpr int PT_STRIDE = NF + NF * NAV + NAV_T;            // F, J, H (packed) of one point (NZ == 0: f (NX), J (NX x NA), H over (x, u))
constexpr int NPT_E = (M == 0) ? 1 : (DENSE_EDGE ? NI * (DEG + 1) : NI * DEG);    // points evaluated per edge (DAE: also point 0 of every element - its algebraic rows)
constexpr int MO_PT = 0;
constexpr int MO_LT = MO_PT + NPT_E * PT_STRIDE;                             // lterm: val, g[NAV], H (packed)
constexpr int MO_MT = MO_LT + 1 + NAV + NAV_T;                               // mterm: val, g[NX], H (packed)
constexpr int MO_NL = MO_MT + 1 + NX + NX_T;                                 // nlcons, per evaluation: d[NEB], Jd[NEB*NAV], H (packed)
constexpr int NL_STRIDE = NEB + NEB * NAV + NAV_T;
constexpr int MO_SIZE = ((MO_NL + NLB * NL_STRIDE + 7) / 8) * 8;
// Compact form of the record (single finite element, continuous model): only the entries that depend on the iterate
// travel through HBM - the generated dompc_*_c functions write them one after the other (lowering.py: compact()); the
// structural zeros and model constants of the dense layout above (149 + 9 of the 231 entries of an industrial_poly
// collocation point, the whole Hessian of its linear stage cost) live in a dense IMAGE of the record that every
// wavefront keeps in its LDS region: initialised once per phase (mo_image_init), the variable entries of the current
// edge scattered into it (mo_expand).  All consumers read the image through the dense indices.
constexpr bool MO_COMPACT = (NI == 1) && (M > 0) && !DENSE_EDGE;
constexpr int MOC_LT = NCOLL * DOMPC_DYN_NV;
constexpr int MOC_MT = MOC_LT + DOMPC_LT_NV;
constexpr int MOC_NL = MOC_MT + DOMPC_MT_NV;
constexpr int MOC_N = MOC_NL + DOMPC_NL_NV;
constexpr int MOC_SIZE = ((MOC_N + 7) / 8) * 8 > 0 ? ((MOC_N + 7) / 8) * 8 : 8;
constexpr int MO_REC = MO_COMPACT ? MOC_SIZE : MO_SIZE;          // doubles per edge in global memory

// per node -------------------------------------------------------------------------------------
constexpr int ND_P = 0;                      // NA x NA
constexpr int ND_PV = ND_P + NA * NA;
constexpr int ND_K = ND_PV + NA;             // NV x NA
constexpr int ND_KV = ND_K + NV * NA;
constexpr int ND_DXT = ND_KV + NV;           // NA
constexpr int ND_AT = ND_DXT + NA;           // FREE_ROOT, used in the root's record: arrival cost value, gradient (NX), Hessian (NX x NX), times the objective scaling
constexpr int AT_LEN = FREE_ROOT ? 1 + NX + NX * NX : 0;
constexpr int ND_SIZE = ((ND_AT + AT_LEN + 7) / 8) * 8;

struct WsLayout {
  int64_t x, zl, zu, lb, ub, dx, gf, rd, xt, dx_sv;
  int64_t lam, dlam, c, ct, dlam_sv;
  int64_t s, zsl, zsu, sl, su, ds, st, ds_sv;
  int64_t ew, es, nd, mo, gsc, total;
  int64_t x_wd, zl_wd, zu_wd, lam_wd, s_wd, zsl_wd, zsu_wd, dlam_e, sgn;      // watchdog: the iterate it started from; EPS_GLOBAL: scratch multiplier steps
};

DOMPC_HD inline WsLayout ws_layout(int n_opt_x, int n_g, int n_edges, int e_pad, int n_nodes) {
  WsLayout L;
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += (n + 7) & ~int64_t(7); return r; };
  L.x = take(n_opt_x); L.zl = take(n_opt_x); L.zu = take(n_opt_x); L.lb = take(n_opt_x); L.ub = take(n_opt_x);
  L.dx = take(n_opt_x); L.gf = take(n_opt_x); L.rd = take(n_opt_x); L.xt = take(n_opt_x);
  L.dx_sv = take(n_opt_x);         // (dx_sv / dlam_sv / ds_sv: the Newton direction while a second-order correction is tried)
  L.lam = take(n_g); L.dlam = take(n_g); L.c = take(n_g); L.ct = take(n_g); L.dlam_sv = take(n_g);
  int64_t nsl = (int64_t)n_edges * NE1;
  L.s = take(nsl); L.zsl = take(nsl); L.zsu = take(nsl); L.sl = take(nsl); L.su = take(nsl);
  L.ds = take(nsl); L.st = take(nsl); L.ds_sv = take(nsl);
  L.ew = take((int64_t)EW_SIZE * e_pad);
  L.es = take((int64_t)ES_SIZE * n_edges);
  L.nd = take((int64_t)ND_SIZE * n_nodes);
  L.mo = take((int64_t)MO_REC * n_edges);
  L.gsc = take(EPS_GLOBAL ? NVG_MAX * (NVG_MAX + 4) : 0);      // shared slack variables: Schur complement, its Cholesky factor, right-hand side / step
  // (touched only while a watchdog is active, solve_problem; its direction is kept in dx_sv / dlam_sv / ds_sv - no second-order correction runs meanwhile)
  L.x_wd = take(n_opt_x); L.zl_wd = take(n_opt_x); L.zu_wd = take(n_opt_x); L.lam_wd = take(n_g);
  L.s_wd = take(nsl); L.zsl_wd = take(nsl); L.zsu_wd = take(nsl);
  L.dlam_e = take(EPS_GLOBAL ? n_g : 0);
  L.sgn = take(nsl);              // scaling factors of the nl_cons rows (IPOPT's gradient-based constraint scaling, solve_problem)
  o += 256;                       // slack: block-granular staging reads of the last records may run past their end
  L.total = o;
  return L;
}

// ------------------------------------------------------------------------------------------------
// LDS pointers carry their address space in the type.  With generic pointers the compiler may fall back to
// FLAT instructions when address-space inference fails, and a flat access whose *base register* is
// (legitimately) a few bytes below the LDS aperture - e.g. &Mx[row][slot*NX + a] with slot = -1 before a
// positive immediate offset is added - faults, while the same ds_read/ds_write is fine.
#ifndef DOMPC_HOST_EMU
typedef __attribute__((address_space(3))) double ldsd;
#else
typedef double ldsd;
#endif

// execution context of the threads that work on one problem: one workgroup, or - "wide" mode, used for
// small batches so that a single make_step can use many CUs - K workgroups that synchronise through a
// device-scope barrier.  All loops over work items are written against (tid, nt), the index / count
// among ALL threads of the problem; (ltid, lnt) are the coordinates inside the workgroup (LDS indexing).
// exchange context of a sharded problem, copied out of the kernel arguments (by value: the argument block
// itself must not escape into out-of-line code, or the compiler loses the address spaces of all its pointers)
struct XCtx {
  int on, rank, world;
  double* xbuf;
  volatile uint32_t *req, *ack, *cnt;
  void (*cb)(void* ctx, double* buf, int32_t count);
  void* ctx;
};
DOMPC_HD inline XCtx make_xctx(const KArgs& A) {
  XCtx X;
  X.on = A.x_mask != nullptr; X.rank = A.shard_rank; X.world = A.shard_world; X.xbuf = A.xbuf;
  X.req = A.x_req; X.ack = A.x_ack; X.cnt = A.x_count; X.cb = A.x_callback; X.ctx = A.x_ctx;
  return X;
}

#ifndef DOMPC_NO_WIDE
#define DOMPC_NO_WIDE 0             // 1: a code object for batch launches only - one workgroup per problem is a compile-time fact (experiment / A/B)
#endif
constexpr bool WIDE_OK = DOMPC_NO_WIDE == 0;
#ifndef DOMPC_BLOCK_CONST
#define DOMPC_BLOCK_CONST 0         // 64: a code object for launches with 64-thread workgroups only (one wavefront per problem: B >= 4096) - thread counts,
#endif                              // strides and the number of lane groups are compile-time constants, workgroup barriers fold away
#if DOMPC_BLOCK_CONST
#define DOMPC_BDIM DOMPC_BLOCK_CONST
#else
#define DOMPC_BDIM ((int)blockDim.x)
#endif
struct Thr {
  int tid, nt;
  ldsd* red;        // LDS: RED_MAX * lnt doubles (the pool is at least that large: `pool` doubles)
  ldsd* filt;       // LDS: 2*MAX_FILTER doubles (every workgroup keeps an identical copy)
  int* flags;       // 8 ints shared by all threads of the problem: LDS (one workgroup) or global (wide)
  ldsd* edge_lds;   // LDS: (lnt/gs) * EL_SIZE doubles (per-group edge working set)
  long long* prof;  // optional sub-phase cycle counters (thread 0 only; may be null)
  int gs;           // lanes cooperating on one edge (64 = one wavefront on the device, 1 in the host emulation)
  int ltid, lnt;    // thread index / count inside the workgroup
  int wg, nwg;      // workgroup index / count of this problem
  unsigned* bar;    // wide: global arrival counter of the problem slot (monotonic)
  double* partials; // wide: global [2][nwg][RED_MAX] reduction partials
  mutable unsigned gen, nred;
  XCtx X;           // tree sharding: exchange buffer and handshake words (X.on == 0: not sharded)
  mutable unsigned xseq;
  const void* kp;   // device: the kernel's argument block in the kernarg segment (handed to the outlined phases)
  // wide mode: every workgroup of the problem runs on the SAME XCD (verified at kernel start from HW_REG_XCC_ID, xcd_census):
  // they share one L2, so the release side of the barrier needs no L2 write-back - the stores only have to have left the CU
  mutable bool light = false;
  int pool = 0;     // doubles in the workgroup's LDS pool (device; KArgs::pool_doubles)
  // two-level barrier (DOMPC_HIER_BARRIER, several XCDs): workgroups of the problem on this XCD / XCDs that hold any (0: flat barrier)
  mutable int xcc = 0, n_local = 0, n_xcd = 0;
  // flags: LDS words in a one-workgroup problem; global words shared by the K workgroups of a wide problem - those are
  // read and written with agent-scope atomics (a plain load could be served from this CU's L1).
  DOMPC_DEV void fset(int i, int v) const {
#ifndef DOMPC_HOST_EMU
    if (WIDE_OK && nwg > 1) { __hip_atomic_store(flags + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
#endif
    flags[i] = v;
  }
  DOMPC_DEV int fget(int i) const {
#ifndef DOMPC_HOST_EMU
    if (WIDE_OK && nwg > 1) return __hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    return flags[i];
  }
  // Failure flag `i` of a phase (0: Riccati pass, 1: sweep): returns the value that means "set" in this phase.  One workgroup: 1, with
  // the reset between two barriers (every thread past its last read of the previous phase, nobody sets before the reset).  Wide mode
  // (round 5): the barrier generation at the phase's entry - no earlier phase used that value, so the flag needs no reset and the
  // phase starts with ONE device-scope barrier instead of two (each costs ~10 us with 100+ workgroups on eight XCDs).
  DOMPC_DEV int flag_begin(int i) const {
    sync();
#ifndef DOMPC_HOST_EMU
    if (WIDE_OK && nwg > 1) return (int)(gen & 0x3fffffffu) + 1;
#endif
    if (tid == 0) fset(i, 0);
    sync();
    return 1;
  }
  DOMPC_DEV void sync() const {
#ifndef DOMPC_HOST_EMU
    if (WIDE_OK && nwg > 1) {
      // device-scope barrier (MI355X_MICROARCH.md, inter-workgroup visibility).  Release side: EVERY wavefront drains
      // its own outstanding global stores (a workgroup-scope barrier does not wait for vmcnt outside tgsplit mode, so
      // without this a peer wavefront's stores could still be in flight when wavefront 0 signals the arrival); after the
      // workgroup barrier lane 0 writes the XCD's L2 back (agent-scope release), drains, and arrives on the monotonic
      // counter.  Acquire side: relaxed polling (bounded), ONE agent-scope acquire (invalidates this CU's L1, which all
      // wavefronts of the workgroup share), workgroup barrier, then plain loads.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      ++gen;
#if DOMPC_HIER_BARRIER
      if (n_xcd > 1) {
        // Two-level barrier for a problem whose workgroups sit on several XCDs (whole-chip wide mode).  Level 1, inside an XCD: the
        // workgroups arrive on a counter that lives in THEIR L2 (workgroup-scope read-modify-write: executed by the L2's atomic
        // unit, never cached in an L1) - every one of them has drained its stores into that L2 before (vmcnt(0) above).  The LAST
        // arrival of the XCD is its leader for this round: ONE L2 write-back per XCD (instead of one per workgroup), arrival on the
        // device-wide counter (8 participants instead of K), spin there, then it releases its XCD through a word in the L2.  The
        // others poll that word with read-modify-writes (L2 round trips instead of trips to memory).  Everybody ends with the
        // agent-scope acquire (its CU's L1; the L2 was invalidated by whoever came first).
        if (ltid == 0) {
          // (its own counters and its own round number: the two barriers of the census ran on the flat counter before this one was set up)
          const unsigned hr = gen - 2u;
          const unsigned old = __hip_atomic_fetch_add(bar + 8 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          long long spins = 0;
          if (old + 1u == hr * (unsigned)n_local) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(bar + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = hr * (unsigned)n_xcd;
            while (__hip_atomic_load(bar + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
              __builtin_amdgcn_s_sleep(2);
              if (++spins > 40000000ll || __hip_atomic_load(flags + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(flags + 7, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_exchange(bar + 16 + xcc, hr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            // (the poll must be a read-modify-write executed by the L2: the compiler turns an atomic add of 0 into an atomic LOAD,
            //  which at workgroup scope is served by this CU's L1 and never sees the leader's write - hence the instruction itself)
            auto poll = [&]() {
              unsigned v;
              asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(bar + 16 + xcc), "v"(0u) : "memory");
              return v;
            };
            while (poll() < hr) {
              __builtin_amdgcn_s_sleep(1);
              if (++spins > 80000000ll || ((spins & 1023) == 0 && __hip_atomic_load(flags + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(flags + 7, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          }
        }
        __syncthreads();
        return;
      }
#endif
      if (ltid == 0) {
        if (!light) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // (buffer_wbl2 sc1: 1.7 - 6.5 us per barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = gen * (unsigned)nwg;
        long long spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > 40000000ll || __hip_atomic_load(flags + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(flags + 7, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // abort: a peer is missing
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
    } else {
      __syncthreads();
    }
#endif
  }
  DOMPC_DEV void lsync() const {        // workgroup-local barrier
#ifndef DOMPC_HOST_EMU
    __syncthreads();
#endif
  }
  // barrier among the lanes of one edge group.  A group is exactly one wavefront on the device and its
  // LDS region is private to it: LDS operations of a wavefront execute in order, so a wavefront-scope
  // fence (keeps the compiler from reordering) is sufficient - no workgroup barrier.
  DOMPC_DEV void gsync() const {
#ifndef DOMPC_HOST_EMU
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
  }
  // Cross-rank exchange of a sharded problem: element-wise SUM over the ranks of X.xbuf[off..off+n).  Called by
  // ALL threads of the problem after they have written their part of the buffer.  Device: workgroup 0
  // publishes the request in pinned host memory and polls the acknowledge word while the host service loop
  // (dompc_runtime.cpp) runs the collective (RCCL all-reduce) on the buffer; bounded spin -> abort flag.
  DOMPC_DEV void xchg(int off, int n) const {
#ifdef DOMPC_HOST_EMU
    if (X.cb) X.cb(X.ctx, X.xbuf + off, n);
#else
    sync();
    ++xseq;
    if (wg == 0 && ltid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      X.cnt[0] = (unsigned)n;
      X.cnt[1] = (unsigned)off;
      __hip_atomic_store((unsigned*)X.req, xseq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      long long spins = 0;
      const bool dead = __hip_atomic_load(flags + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      while (!dead && __hip_atomic_load((unsigned*)X.ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != xseq) {
        __builtin_amdgcn_s_sleep(16);
        if (++spins > 4000000ll) {                        // (seconds) the host never answered: abort instead of hanging
          __hip_atomic_store(flags + 7, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    sync();
#endif
  }
};

#ifndef DOMPC_HOST_EMU
// LDS of a workgroup (file scope: the outlined phase functions below rebuild their thread context from it).
// One pool (dynamic: the runtime sizes it for the wavefronts per workgroup it launches): per-wavefront edge / node working
// sets during sweeps and Riccati passes, reduction scratch otherwise (never live at the same time; every use is
// bracketed by workgroup barriers).
extern __shared__ double lds_pool[];
__shared__ double lds_filt[2 * MAX_FILTER];
__shared__ int lds_flags[8];
__shared__ int lds_b;
#ifndef DOMPC_HIER_BARRIER
#define DOMPC_HIER_BARRIER 0        // 1: two-level barrier (per-XCD counters in the L2, one write-back and one device-wide arrival per XCD) when the workgroups of a problem sit on several XCDs
#endif
#ifndef DOMPC_LIGHT_BARRIER
#define DOMPC_LIGHT_BARRIER 1       // wide mode: barrier without the L2 write-back when the problem's workgroups share an XCD (0: always write back)
#endif
#ifndef DOMPC_PROFILE
#define DOMPC_PROFILE 0             // 1: sub-phase shader-clock counters of the edge sweep / node update (tools/gpu_profile.py)
#endif
__shared__ long long lds_prof[32];

// slot of the calling workgroup: normal mode one workgroup per problem slot; wide mode (small batches) K = A.wide
// workgroups per problem, all on one XCD when the dispatcher places block b on XCD b % 8 (affinity only - the barrier
// protocol does not depend on it): b = 8*q + r, workgroup-in-problem j = q % K, slot = (q / K) * 8 + r.
__device__ inline int slot_of_block(const KArgs& A) {
  const bool wide = (A.mode == 0 && A.wide > 1);
  const int q = blockIdx.x / 8;
  if (wide && A.wide_spread) return (int)blockIdx.x / A.wide;      // whole-chip placement: the K workgroups of a problem are consecutive blocks (all XCDs)
  return wide ? (q / A.wide) * 8 + (int)(blockIdx.x % 8) : (int)blockIdx.x;
}
__device__ inline Thr make_thr(const KArgs& A) {
  const bool wide = WIDE_OK && (A.mode == 0 && A.wide > 1);
  const int K = wide ? A.wide : 1;
  const int q = blockIdx.x / 8;
  const int j = wide ? (A.wide_spread ? (int)blockIdx.x % K : q % K) : 0;
  const int slot = slot_of_block(A);
  return Thr{j * DOMPC_BDIM + (int)threadIdx.x, K * DOMPC_BDIM, (ldsd*)lds_pool, (ldsd*)lds_filt,
             wide ? A.wide_flags + slot * 8 : lds_flags, (ldsd*)lds_pool, lds_prof, 64,
             (int)threadIdx.x, DOMPC_BDIM, j, K, wide ? A.wide_bar + slot * WIDE_BAR_STRIDE : nullptr,
             wide ? A.wide_partials + (int64_t)slot * 2 * K * RED_MAX : nullptr, 0u, 0u, make_xctx(A), 0u, nullptr,
             (wide && DOMPC_LIGHT_BARRIER) ? __hip_atomic_load(A.wide_bar + slot * WIDE_BAR_STRIDE + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u : false,
             A.pool_doubles};
}
// wide mode, once per launch: do all workgroups of this problem run on one XCD?  Every workgroup publishes its XCC id
// (hardware register), the first workgroup writes the verdict (word 2 of the slot's barrier block: 1 = one XCD, 2 = several)
// between two full barriers; make_thr of the outlined phases reads it back.  Placement is NOT assumed (the dispatcher puts
// block b on XCD b % 8 today, slot_of_block): on any other placement the barrier keeps its L2 write-back.
// two-level barrier: this workgroup's XCD, the number of the problem's workgroups on it and the number of XCDs that hold any - from the
// census words (valid after the census of the launch; the outlined phases rebuild their Thr and read them again)
__device__ inline void hier_setup(const Thr& T) {
#if DOMPC_HIER_BARRIER
  if (!WIDE_OK || T.nwg <= 1 || T.light) { T.n_xcd = 0; return; }
  T.xcc = (int)((unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);
  const unsigned m = __hip_atomic_load(T.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  T.n_xcd = __builtin_popcount(m & 0xffu);
  T.n_local = (int)__hip_atomic_load(T.bar + 24 + T.xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (T.n_xcd <= 1 || T.n_local <= 0 || (m >> 8)) T.n_xcd = 0;         // (one XCD: the light barrier; an XCC id above 7: flat barrier)
#else
  (void)T;
#endif
}
__device__ inline void xcd_census(const Thr& T) {
  if (!WIDE_OK || T.nwg <= 1 || !DOMPC_LIGHT_BARRIER) return;
  if (T.ltid == 0) {
    const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;          // HW_REG_XCC_ID[3:0]
    __hip_atomic_fetch_or(T.bar + 1, 1u << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (DOMPC_HIER_BARRIER) __hip_atomic_fetch_add(T.bar + 24 + (xcc & 7u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  T.sync();
  if (T.tid == 0) {
    const unsigned m = __hip_atomic_load(T.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(T.bar + 2, (m & (m - 1u)) == 0u ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  T.sync();
  T.light = __hip_atomic_load(T.bar + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u;
  hier_setup(T);
}
// wave-uniform copies of values that reach an outlined function in vector registers
__device__ inline int ufl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline unsigned ufl(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline double ufl(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = ufl((unsigned)u), hi = ufl((unsigned)(u >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
// The kernel's argument block, read through the kernarg-segment pointer the kernel obtained and passed down (the
// pointer builtin itself is only valid inside the kernel function): made uniform -> scalar loads.
typedef const __attribute__((address_space(4))) KArgs* KArgsCP;
__device__ inline KArgs kernel_args(const void* kp) {
  const unsigned long long u = (unsigned long long)kp;
  const unsigned lo = ufl((unsigned)u), hi = ufl((unsigned)(u >> 32));
  return *(KArgsCP)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}
#endif

// ---- tree sharding: masks 0 = another rank's, 1 = mine, 2 = replicated (identical everywhere; counted once).
// The support is compiled in only with -DDOMPC_SHARD=1 (a second code object per model, build.py): in the
// plain build every helper folds to a constant and the batch path carries no mask loads and no extra registers
// (measured: the mask-aware build is 9 % slower on the B=1024 batch).
constexpr bool SHARD = DOMPC_SHARD != 0;
DOMPC_DEV inline bool sh_on(const KArgs& A) { return SHARD && A.x_mask != nullptr; }
DOMPC_DEV inline int mk_x(const KArgs& A, int g) { return (SHARD && A.x_mask) ? A.x_mask[g] : 1; }
DOMPC_DEV inline int mk_g(const KArgs& A, int r) { return (SHARD && A.g_mask) ? A.g_mask[r] : 1; }
DOMPC_DEV inline int mk_e(const KArgs& A, int e) { return (SHARD && A.e_mask) ? A.e_mask[e] : 1; }
DOMPC_DEV inline int mk_n(const KArgs& A, int n) { return (SHARD && A.n_mask) ? A.n_mask[n] : 1; }
// does an item with mask m enter a SUM on this rank?
DOMPC_DEV inline bool sh_cnt(const KArgs& A, int m) { return !SHARD || m == 1 || (m == 2 && A.shard_rank == 0); }
DOMPC_DEV inline int cut_of(const KArgs& A, int n) { return (SHARD && A.node_cut) ? A.node_cut[n] : -1; }


DOMPC_DEV inline long long prof_clock() {
#ifndef DOMPC_HOST_EMU
  return (long long)clock64();
#else
  return 0;
#endif
}

// stop request of the host (watchdog of the blocking entry points, dompc_abort): system-scope load of the pinned word
DOMPC_DEV inline int abort_requested(const KArgs& A) {
#ifndef DOMPC_HOST_EMU
  return A.abort_flag ? __hip_atomic_load(A.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
#else
  return A.abort_flag ? *(const volatile int32_t*)A.abort_flag : 0;
#endif
}

enum RedOp { R_SUM = 0, R_MAX = 1, R_MIN = 2 };

// Reduce n values per thread across the workgroup; result broadcast to every thread.
template <int N_>
DOMPC_DEV void wg_reduce(const Thr& T, double (&v)[N_], const int (&op)[N_]) {
  static_assert(N_ <= RED_MAX, "too many values");
  if (T.nt == 1 && !(SHARD && T.X.on)) return;
  for (int i = 0; i < N_; ++i) T.red[i * T.lnt + T.ltid] = v[i];
  T.lsync();
  for (int s = T.lnt >> 1; s > 0; s >>= 1) {
    if (T.ltid < s) {
      for (int i = 0; i < N_; ++i) {
        double a = T.red[i * T.lnt + T.ltid], b = T.red[i * T.lnt + T.ltid + s];
        T.red[i * T.lnt + T.ltid] = op[i] == R_SUM ? a + b : (op[i] == R_MAX ? fmax(a, b) : fmin(a, b));
      }
    }
    T.lsync();
  }
  if (T.nwg > 1) {
    // combine the workgroups' partials in a fixed order (bitwise identical on every workgroup)
    double* buf = T.partials + (T.nred & 1u) * T.nwg * RED_MAX;
    ++T.nred;
    if (T.ltid < N_) buf[T.wg * RED_MAX + T.ltid] = T.red[T.ltid * T.lnt];
    T.sync();
#ifndef DOMPC_HOST_EMU
    // Whole-chip wide mode (round 5: up to 256 workgroups per problem on all XCDs): the rows of the other workgroups sit in memory behind
    // the barrier's L2 invalidate, and folding them one dependent-looking load after the other cost ~1 us per row (a quarter of an IPM
    // iteration of the 243-leaf tree went into these folds).  All threads of the workgroup fetch the table at once into the free part
    // of the LDS pool (behind the reduction scratch), then the N_ folding threads read it from there - in the same order: same bits.
    const int n_tab = T.nwg * RED_MAX;
    ldsd* tab = T.red + RED_MAX * T.lnt;
    const bool staged = RED_MAX * T.lnt + n_tab <= T.pool;
    if (staged) {
      for (int i = T.ltid; i < n_tab; i += T.lnt) tab[i] = buf[i];
      T.lsync();
    }
#else
    const bool staged = false;
    const double* tab = nullptr;
#endif
    if (T.ltid < N_) {
      double acc = staged ? (double)tab[T.ltid] : buf[T.ltid];
      for (int w = 1; w < T.nwg; ++w) {
        const double b = staged ? (double)tab[w * RED_MAX + T.ltid] : buf[w * RED_MAX + T.ltid];
        acc = op[T.ltid] == R_SUM ? acc + b : (op[T.ltid] == R_MAX ? fmax(acc, b) : fmin(acc, b));
      }
      T.red[T.ltid * T.lnt] = acc;
    }
    T.lsync();
  }
  if (SHARD && T.X.on) {
    // combine the ranks: every rank deposits its values in its own row of a [world][RED_MAX] table (zeros
    // elsewhere), the SUM exchange turns that into an all-gather, and every rank folds the rows in rank order
    // with the requested operations -> bitwise identical results and control flow on all ranks
    const int W = T.X.world, me = T.X.rank;
    double* xb = T.X.xbuf;
    if (T.wg == 0)
      for (int i = T.ltid; i < W * RED_MAX; i += T.lnt) {
        const int w = i / RED_MAX, j = i % RED_MAX;
        xb[i] = (w == me && j < N_) ? T.red[j * T.lnt] : 0.0;
      }
    T.xchg(0, W * RED_MAX);
    for (int i = T.ltid; i < N_; i += T.lnt) {
      double acc = xb[i];
      for (int w = 1; w < W; ++w) {
        const double b = xb[w * RED_MAX + i];
        acc = op[i] == R_SUM ? acc + b : (op[i] == R_MAX ? fmax(acc, b) : fmin(acc, b));
      }
      T.red[i * T.lnt] = acc;
    }
    T.lsync();
  }
  for (int i = 0; i < N_; ++i) v[i] = T.red[i * T.lnt];
  T.lsync();
}

// ------------------------------------------------------------------------------------------------
// Small dense product for one lane group:  D (m x n, row-major, ldd) = beta*D + op(A) (m x k) * op(B) (k x n)
// with A(i,l) = A[i*sai + l*sal], B(l,j) = B[l*sbl + j*sbj].  Every lane of the group must call it; the
// caller separates it from producers/consumers of the operands with gsync().
// Device: the FP64 matrix cores, v_mfma_f64_16x16x4_f64 on 16x16 tiles with zero padding (the stage blocks
// are 13x13 / 10x13 / 13x30: one or two tiles) - operand fragment A[l&15][4kb+(l>>4)], B[4kb+(l>>4)][l&15],
// result col = l&15, row = (l>>4) + 4*reg.  Host emulation: plain loops.
DOMPC_DEV inline void gmm(int lane, int GS, int m, int n, int k, const double* A, int sai, int sal,
                          const double* B, int sbl, int sbj, double beta, double* D, int ldd) {
#ifndef DOMPC_HOST_EMU
  typedef double d4 __attribute__((ext_vector_type(4)));
  (void)GS;
  const int li = lane & 15, lk = lane >> 4;
  for (int ti = 0; ti < m; ti += 16)
    for (int tj = 0; tj < n; tj += 16) {
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      const int ai = ti + li, bj = tj + li;
      for (int kb = 0; kb < k; kb += 4) {
        const int kk = kb + lk;
        const double a = (ai < m && kk < k) ? A[ai * sai + kk * sal] : 0.0;
        const double b = (bj < n && kk < k) ? B[kk * sbl + bj * sbj] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
      const int col = tj + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = ti + lk + 4 * r;
        if (row < m && col < n) {
          double* d = D + row * ldd + col;
          *d = (beta == 0.0) ? acc[r] : beta * (*d) + acc[r];
        }
      }
    }
#else
  for (int it = lane; it < m * n; it += GS) {
    const int i = it / n, j = it % n;
    double t = 0.0;
    for (int l = 0; l < k; ++l) t += A[i * sai + l * sal] * B[l * sbl + j * sbj];
    D[i * ldd + j] = (beta == 0.0) ? t : beta * D[i * ldd + j] + t;
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// per-problem view
struct Prob {
  const KArgs* A;
  const double* P;                                   // opt_p of this problem
  double *x, *zl, *zu, *lb, *ub, *dx, *gf, *rd, *xt, *dx_sv;
  double *lb_own, *ub_own;                           // this slot's copies of the bounds (lb / ub: the ones the phases read, prob_bounds)
  double *lam, *dlam, *c, *ct, *dlam_sv;
  double *s, *zsl, *zsu, *sl, *su, *ds, *st, *ds_sv;
  double *ew, *es, *nd, *mo, *gsc;
  double *x_wd, *zl_wd, *zu_wd, *lam_wd, *s_wd, *zsl_wd, *zsu_wd, *dlam_e, *sgn;
  int e_pad;
  double sf;                                         // objective scaling
  double mu;
  int soc;                                           // bit 0: the constraint residual c is an INPUT of the sweep (second-order correction)
                                                     // bit 1: the Hessians of the objective terms (lterm, mterm, rterm) are left out -
                                                     //        with lambda = 0, z = 0 and delta = dsw = 1 the system of the
                                                     //        least-squares multiplier estimate [I A'; A 0] (solve_problem)
  double dsw;                                        // inertia correction that the sweep has already folded into the condensed blocks
                                                     // (Sigma_w + dsw in Q~, q~ and in the stored Sigma_w): the Riccati passes add
                                                     // only delta - dsw on the eliminated variables (0 in the common case)
  int slot;                                          // workspace slot of the problem (rebuilds this view inside outlined functions)
  int rp_soc; double rp_mu;                          // DOMPC_REPEAT_PHASE (measurement builds): arguments of the last sweep
  int lu_ok;                                         // QUAD_FWD: the forward records hold G_cc^-1 (the last sweep stored it: lu_store_rule)
  DOMPC_DEV double& EW(int e, int i) const { return ew[(int64_t)e * EW_SIZE + i]; }
  DOMPC_DEV double* ES(int e) const { return es + (int64_t)e * ES_SIZE; }
  DOMPC_DEV double* ND(int n) const { return nd + (int64_t)n * ND_SIZE; }
  DOMPC_DEV double* MO(int e) const { return mo + (int64_t)e * MO_REC; }
};

DOMPC_DEV inline Prob make_prob(const KArgs& A, int slot, const double* P) {
  WsLayout L = ws_layout(A.n_opt_x, A.n_g, A.n_edges, A.e_pad, A.n_nodes);
  double* w = A.ws + (int64_t)slot * A.ws_stride;
  Prob p;
  p.A = &A; p.P = P;
  p.x = w + L.x; p.zl = w + L.zl; p.zu = w + L.zu; p.lb_own = w + L.lb; p.ub_own = w + L.ub; p.dx = w + L.dx;
  p.lb = A.lb_sh ? A.lb_sh : p.lb_own; p.ub = A.ub_sh ? A.ub_sh : p.ub_own;
  p.gf = w + L.gf; p.rd = w + L.rd; p.xt = w + L.xt; p.dx_sv = w + L.dx_sv;
  p.lam = w + L.lam; p.dlam = w + L.dlam; p.c = w + L.c; p.ct = w + L.ct; p.dlam_sv = w + L.dlam_sv;
  p.s = w + L.s; p.zsl = w + L.zsl; p.zsu = w + L.zsu; p.sl = w + L.sl; p.su = w + L.su;
  p.ds = w + L.ds; p.st = w + L.st; p.ds_sv = w + L.ds_sv;
  p.ew = w + L.ew; p.es = w + L.es; p.nd = w + L.nd; p.mo = w + L.mo; p.gsc = w + L.gsc;
  p.x_wd = w + L.x_wd; p.zl_wd = w + L.zl_wd; p.zu_wd = w + L.zu_wd; p.lam_wd = w + L.lam_wd;
  p.s_wd = w + L.s_wd; p.zsl_wd = w + L.zsl_wd; p.zsu_wd = w + L.zsu_wd; p.dlam_e = w + L.dlam_e; p.sgn = w + L.sgn;
  p.e_pad = A.e_pad; p.sf = 1.0; p.mu = 0.0; p.soc = 0; p.dsw = 0.0; p.slot = slot; p.lu_ok = 1; p.rp_soc = 0; p.rp_mu = 0.0;
  return p;
}
// Bounds the phases read: ONE copy for all problems of the launch (KArgs::lb_sh / ub_sh) - except while the least-squares multiplier
// estimate of a problem runs (Prob::soc bit 1): its sweep works on bounds one unit away from the problem's own starting point, kept
// in the slot's copies.
DOMPC_DEV inline void prob_bounds(Prob& Q) {
  const KArgs& A = *Q.A;
  const bool own = (Q.soc & 2) != 0 || !A.lb_sh;
  Q.lb = own ? Q.lb_own : A.lb_sh;
  Q.ub = own ? Q.ub_own : A.ub_sh;
}

// slot of collocation point r of finite element i inside the edge's w block (optimizer.py:905-935)
DOMPC_DEV constexpr int slot_of(int i, int r) { return i == 0 ? r - 1 : DEG + (i - 1) * (DEG + 1) + r; }
DOMPC_DEV constexpr int next_slot(int i) { return (i + 1 < NI) ? slot_of(i + 1, 0) : M - 1; }

// Index of the lane group (= wavefront on the device) a thread belongs to, as a wave-uniform value: everything derived
// from it - the edge / node number, the table look-ups, the record pointers - then lives in SGPRs (scalar loads, SGPR
// base + 32-bit lane offset addressing) instead of one 64-bit VGPR address pair and one vector load per look-up.
DOMPC_DEV inline int group_index(int tid, int gs) {
#ifndef DOMPC_HOST_EMU
  return __builtin_amdgcn_readfirstlane(tid / gs);
#else
  return tid / gs;
#endif
}

// base[idx] with the BYTE offset formed in 32-bit arithmetic: wave-uniform base (SGPR pair) + zero-extended 32-bit lane
// offset is an addressing mode of the global loads; an index that is scaled after its extension to 64 bits is not.
DOMPC_DEV inline double ldoff(const double* base, unsigned idx) {
#ifndef DOMPC_HOST_EMU
  return *(const double*)((const char*)base + (idx << 3));
#else
  return base[idx];
#endif
}

// Entry of a small compile-time table (collocation coefficients, input scalings, rterm weights) at a LANE-DEPENDENT index,
// entries [first, first + count): selects over values the optimiser cannot see through (an empty asm per entry).  An
// indexed read is a vector load from constant memory + vmcnt(0) in the middle of a phase - a memory round trip that also
// waits for every prefetch and store in flight (two of them were 30 % of the Riccati node update) - and a plain select
// chain over literals is folded straight back into such a lookup-table load.
template <int N>
DOMPC_DEV inline double tab_sel(const double (&tab)[N], int idx, int first = 0, int count = N) {
#ifndef DOMPC_HOST_EMU
  double v = tab[first];
  asm("" : "+v"(v));
#pragma unroll
  for (int i = 1; i < N; ++i) {
    if (i < count) {
      double t = tab[first + i];
      asm("" : "+v"(t));
      v = (idx == first + i) ? t : v;
    }
  }
  return v;
#else
  (void)first; (void)count;
  return tab[idx];
#endif
}

// reciprocal of a normal, non-zero double: v_rcp_f64 + two Newton steps (5 instructions instead of the ~12 of the IEEE
// division sequence; the result is within an ulp or two, no denormal / infinity handling - the callers exclude those)
DOMPC_DEV inline double fast_rcp(double x) {
#ifndef DOMPC_HOST_EMU
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}

// IPOPT's linear damping of the barrier function for variables with ONE bound (kappa_d, section 3.7 of the implementation
// paper): phi_mu gets + kappa_d mu (x - l) per lower-only and + kappa_d mu (u - x) per upper-only variable, the primal-dual
// equations and the error measures the gradient of it.  IPOPT's default is 1e-5; with it the oracle reproduces the CSTR and
// batch_reactor goldens to 1e-13 instead of 1e-7 / 2e-11.  Compile-time constant (-DDOMPC_KAPPA_D=0
// builds the kernels without it: every use is guarded, the code is then identical to the one before the term existed).
#ifndef DOMPC_KAPPA_D
#define DOMPC_KAPPA_D 1e-5
#endif
constexpr double KAPPA_D = DOMPC_KAPPA_D;
// +1: lower bound only, -1: upper bound only, 0: none or both
DOMPC_DEV inline double one_sided(double l, double u) {
  const bool hl = l > -INFINITY, hu = u < INFINITY;
  return (hl && !hu) ? 1.0 : ((hu && !hl) ? -1.0 : 0.0);
}
// `damp` = false: without the damping term - the least-squares multiplier estimate of the starting point (solve_problem,
// Prob::soc bit 1) uses the barrier gradient at mu = 1 with the bounds one unit away as a stand-in for -z_L + z_U = -1 + 1;
// IPOPT's estimate has no damping term
DOMPC_DEV inline double bar_grad(double x, double l, double u, double mu, bool damp = true) {
  double g = 0.0;
  if (l > -INFINITY) g -= mu * fast_rcp(x - l);
  if (u < INFINITY) g += mu * fast_rcp(u - x);
  if (KAPPA_D != 0.0 && damp) g += KAPPA_D * mu * one_sided(l, u);
  return g;
}
DOMPC_DEV inline double sigma_of(double x, double l, double u, double zl, double zu) {
  double sg = 0.0;
  if (l > -INFINITY) sg += zl * fast_rcp(x - l);
  if (u < INFINITY) sg += zu * fast_rcp(u - x);
  return sg;
}

DOMPC_DEV inline double edge_rterm_f(const Prob& Q, int e, const double* xv);      // (user-defined rterm, defined below)
DOMPC_DEV inline void edge_rterm_eval(const Prob& Q, int e, ldsd* dst);
DOMPC_DEV inline void edge_rterm_store(const ldsd* src, double* S_, int lane, int GS);
// Node-local cost terms added to `nlp_obj` on the route prepare_nlp -> modify -> create_nlp (optimizer.py:82-129; do_mpc_amd/nlp_route.py
// ObjectiveExtras, lowering.py `extras`): the generated header then defines DOMPC_XTRA, the switch-dispatched functions dompc_xtra_lt /
// dompc_xtra_mt (value-only `_f`, compact `_c`, dense) and the tables edge -> function index.  They ADD to the stage-cost record of the
// node's first outgoing edge / to the terminal-cost record of a leaf's incoming edge (already divided by the edge's omega), so every
// consumer of those records - condensing, gradient, objective value, objective scaling - sees them without knowing.  P: the whole opt_p.
#ifndef DOMPC_XTRA
#define DOMPC_XTRA 0
#endif
// ... and terms in the collocation states `_x[k, s, c]`, c < M, of ONE interval (the reference's docstring example, optimizer.py:91-97, on a
// continuous model): DOMPC_XTRA_EW, dompc_xtra_ew_f / dompc_xtra_ew over the interval's unknowns w.  The generated header then also sets
// DOMPC_FORCE_DENSE: only the dense edge path (dompc_dae.h) carries an objective gradient and Hessian over the edge's own unknowns.
#ifndef DOMPC_XTRA_EW
#define DOMPC_XTRA_EW 0
#endif
// Inequality rows appended to `nlp_cons` that stay inside one node (optimizer.py:131-215; nlp_route.ConstraintExtras): DOMPC_NE counts
// DOMPC_XROW_SLOTS extra row slots behind the DOMPC_XROW_BASE nl_cons rows of every edge; the generated nl_cons functions return zeros
// there and dompc_xrow* add the row that edge e has in a slot (DOMPC_XROW_ID[e * slots + slot]; 0: masked - an identically-zero row with an
// unbounded slack, inert in every formula; DOMPC_XROW_MASKED of them are subtracted from the row count in the driver's s_d).
#ifndef DOMPC_XROW
#define DOMPC_XROW 0
#define DOMPC_XROW_MASKED 0
#endif
DOMPC_DEV inline void nlcons_f_e(const Prob& Q, int e, const double* xs, const double* us, const double* zs, const double* tvp, const double* pp, double* d) {
  dompc_nlcons_f(xs, us, zs, tvp, pp, d);
#if DOMPC_XROW
  for (int r = 0; r < DOMPC_XROW_SLOTS; ++r) d[DOMPC_XROW_BASE + r] += dompc_xrow_f(DOMPC_XROW_ID[e * DOMPC_XROW_SLOTS + r], xs, us, Q.P);
#endif
  (void)e; (void)Q;
}
DOMPC_DEV inline void nlcons_e(const Prob& Q, int e, const double* xs, const double* us, const double* zs, const double* tvp, const double* pp,
                               const double* lam, double* d, double* Jd, double* H) {
  dompc_nlcons(xs, us, zs, tvp, pp, lam, d, Jd, H);
#if DOMPC_XROW
  for (int r = 0; r < DOMPC_XROW_SLOTS; ++r) dompc_xrow(DOMPC_XROW_ID[e * DOMPC_XROW_SLOTS + r], xs, us, Q.P, lam, d, Jd, H);
#endif
  (void)e; (void)Q;
}
DOMPC_DEV inline double lterm_f_e(const Prob& Q, int e, const double* xs, const double* us, const double* zs, const double* tvp, const double* pp) {
  double v = dompc_lterm_f(xs, us, zs, tvp, pp);
#if DOMPC_XTRA
  v += dompc_xtra_lt_f(DOMPC_XTRA_LT_ID[e], xs, us, Q.P);
#endif
  (void)e; (void)Q;
  return v;
}
DOMPC_DEV inline double mterm_f_e(const Prob& Q, int e, const double* xs, const double* tvp, const double* pp) {
  double v = dompc_mterm_f(xs, tvp, pp);
#if DOMPC_XTRA
  v += dompc_xtra_mt_f(DOMPC_XTRA_MT_ID[e], xs, Q.P);
#endif
  (void)e; (void)Q;
  return v;
}
DOMPC_DEV inline void lterm_e(const Prob& Q, int e, const double* xs, const double* us, const double* zs, const double* tvp, const double* pp,
                              double* val, double* g, double* H) {
  dompc_lterm(xs, us, zs, tvp, pp, val, g, H);
#if DOMPC_XTRA
  dompc_xtra_lt(DOMPC_XTRA_LT_ID[e], xs, us, Q.P, val, g, H);
#endif
  (void)e; (void)Q;
}
DOMPC_DEV inline void mterm_e(const Prob& Q, int e, const double* xs, const double* tvp, const double* pp, double* val, double* g, double* H) {
  dompc_mterm(xs, tvp, pp, val, g, H);
#if DOMPC_XTRA
  dompc_xtra_mt(DOMPC_XTRA_MT_ID[e], xs, Q.P, val, g, H);
#endif
  (void)e; (void)Q;
}

#include "dompc_dae.h"       // edge phases of models with algebraic states (dense path)

#include "dompc_edge.h"
#include "dompc_factor.h"
#include "dompc_quad.h"
#include "dompc_node.h"
#include "dompc_riccati.h"
#include "dompc_forward.h"
#include "dompc_sweep.h"
#include "dompc_phases.h"
#include "dompc_driver.h"
}  // namespace dompc
