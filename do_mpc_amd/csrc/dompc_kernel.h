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
#pragma once
#include <math.h>
#include <stdint.h>
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
  p.e_pad = A.e_pad; p.sf = 1.0; p.mu = 0.0; p.soc = 0; p.dsw = 0.0; p.slot = slot;
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
#include "dompc_dae.h"       // edge phases of models with algebraic states (dense path)

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
  if (M == 0) {
    dompc_dyn_f(xn, un, nullptr, tvp, pp, f);
    for (int a = 0; a < NX; ++a) cv[row0 + a] = f[a] - xc[a];
  } else {
    (void)REST;
    for (int i = 0; i < NI; ++i) {
      const double* xi0 = (i == 0) ? xn : w + slot_of(i, 0) * NX;
      const int rb = row0 + i * (DEG + 1) * NX;
      for (int j = 1; j <= DEG; ++j) {
        if (SPLIT && part != i * DEG + (j - 1)) continue;
        const double* xij = w + slot_of(i, j) * NX;
        dompc_dyn_f(xij, un, nullptr, tvp, pp, f);
        for (int a = 0; a < NX; ++a) {
          double xp = DOMPC_C[0 * (DEG + 1) + j] * xi0[a];
          for (int r = 1; r <= DEG; ++r) xp += DOMPC_C[r * (DEG + 1) + j] * w[slot_of(i, r) * NX + a];
          cv[rb + (j - 1) * NX + a] = f[a] - xp;
        }
      }
      if (SPLIT && part != REST) continue;
      const double* xnext = w + next_slot(i) * NX;
      for (int a = 0; a < NX; ++a) {
        double xf = DOMPC_D[0] * xi0[a];
        for (int r = 1; r <= DEG; ++r) xf += DOMPC_D[r] * w[slot_of(i, r) * NX + a];
        cv[rb + DEG * NX + a] = xnext[a] - xf;
      }
    }
    if (SPLIT && part != REST) return 0.0;
    for (int a = 0; a < NX; ++a) cv[row0 + NW + a] = w[(M - 1) * NX + a] - xc[a];
  }
  double obj = om * dompc_lterm_f(xn, un, nullptr, tvp, pp);
  if (k == A.N - 1) obj += om * dompc_mterm_f(xc, Q.P + A.p_off_tvp + (k + 1) * NTVP, pp);
  if (RT_CUSTOM) obj += edge_rterm_f(Q, e, xv);
  if (NE > 0) {
    double d[NE1];
    dompc_nlcons_f(xn, un, nullptr, tvp, pp, d);
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
constexpr int EL_SIZE = ((el_max(el_max(el_max(EL_MOC + MOC_STAGE, RB_NEED), el_max(RF_IMG + MO_IMG, R16_NEED)), DAE_NEED) + 7) / 8) * 8;

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
        dompc_dyn_c(w + slot_of(0, jj) * NX, un, nullptr, tvp, pp, Q.lam + row0 + (jj - 1) * NX, mo + j * DOMPC_DYN_NV);
      } else if (kind == 1) {
        dompc_lterm_c(xn, un, nullptr, tvp, pp, mo + MOC_LT);
      } else if (kind == 2) {
        if (k == A.N - 1) dompc_mterm_c(Q.x + A.node_x_off[cn], Q.P + A.p_off_tvp + (k + 1) * NTVP, pp, mo + MOC_MT);
      } else if (NE > 0) {
        double yds[NE1];      // (scaled rows sg d(x): the Hessian sum_i lambda_i sg_i hess d_i)
        for (int i = 0; i < NE; ++i) yds[i] = Q.lam[row0 + NW + NX + i] * Q.sgn[e * NE1 + i];
        dompc_nlcons_c(xn, un, nullptr, tvp, pp, yds, mo + MOC_NL);
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
      dompc_lterm(xn, un, nullptr, tvp, pp, mo + MO_LT, mo + MO_LT + 1, mo + MO_LT + 1 + NA);
    } else if (kind == 2) {
      if (k == A.N - 1)
        dompc_mterm(Q.x + A.node_x_off[cn], Q.P + A.p_off_tvp + (k + 1) * NTVP, pp, mo + MO_MT, mo + MO_MT + 1,
                    mo + MO_MT + 1 + NX);
    } else if (NE > 0) {
      double yds[NE1];
      for (int i = 0; i < NE; ++i) yds[i] = Q.lam[row0 + NW + NX + i] * Q.sgn[e * NE1 + i];
      dompc_nlcons(xn, un, nullptr, tvp, pp, yds, mo + MO_NL, mo + MO_NL + NE, mo + MO_NL + NE + NE * NA);
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

#ifndef DOMPC_HOST_EMU
// ================================================================================================
// Blocked Gauss-Jordan of the collocation block on the FP64 matrix cores (round 3).
// The register-resident elimination below (one extended column per lane, the pivot column broadcast with v_readlane) issues
// ~60 vector instructions per pivot - two thirds of them broadcasts - and was the largest single phase of the solve (27 %).
// Here the extended matrix  [G_cc (padded to a multiple of 4) | G_y r | I]  lives in 16x16 tiles in the accumulator layout
// of v_mfma_f64_16x16x4_f64 (lane l, register r: element ((l >> 4) + 4 r, l & 15) of the tile) and FOUR pivots are
// eliminated per step with rank-4 updates:
//     P  = A[panel rows, panel cols]  (4 x 4),      C~ = A[:, panel cols] - E_panel   (E_panel: unit rows of the panel),
//     A <- A - (C~ P^-1) A[panel rows, :]           (non-panel rows: A - C P^-1 R; panel rows: P^-1 R)
// - the panel ROWS are register (p % 4) of the tiles of tile row p / 4, i.e. already the B operand of the instruction;
// - the panel COLUMNS go through LDS once per step (20 x 4 doubles): every lane reads P (broadcast reads), factorises it in
//   uniform arithmetic (LU without pivoting, threshold test on its pivots), solves for ITS column k = l >> 4 of P^-1 and
//   forms its entries (row l & 15 of each tile row, column k) of C~ P^-1 - the A operand;
// - 8 (later 6) MFMAs per step instead of ~240 vector instructions for the same four pivots.
// Natural pivot order (the diagonal of G_cc = h J - C (x) I carries the collocation coefficients); a failed threshold test
// returns 1 and the caller repeats the factorisation with the register-resident elimination and partial pivoting.
// Out: W | w0 (collocation rows; the caller derives the continuity rows) in LDS, G_cc^-1 in the forward record.
#ifndef DOMPC_MFMA_GJ
#define DOMPC_MFMA_GJ 1
#endif
#ifndef DOMPC_GJ_SKIP
#define DOMPC_GJ_SKIP 1             // blocked elimination: skip the updates of tile columns whose unit columns are still untouched (0: update everything)
#endif
#ifndef DOMPC_DUAL_VALU
#define DOMPC_DUAL_VALU 1           // dual-residual products of the factorisation on the vector ALU (0: on the matrix cores, multipliers in one row of the A operand)
#endif
#ifndef DOMPC_GJ_PRIO
#define DOMPC_GJ_PRIO 3             // wavefront priority (s_setprio) while the factorisation of an edge runs: its dependent chains then win the
                                    // issue arbitration against the partner wavefront's memory instructions (+1.3 %); 0: off
#endif
#ifndef DOMPC_MM_PRIO
#define DOMPC_MM_PRIO 0             // ... while the tile condensing of the sweep / the matrix part of a Riccati node runs (measured: nothing on top)
#endif
#if DOMPC_MM_PRIO && !defined(DOMPC_HOST_EMU)
#define DOMPC_PRIO_UP() __builtin_amdgcn_s_setprio(DOMPC_MM_PRIO)
#define DOMPC_PRIO_DOWN() __builtin_amdgcn_s_setprio(0)
#else
#define DOMPC_PRIO_UP()
#define DOMPC_PRIO_DOWN()
#endif
#ifndef DOMPC_GJ_U
#define DOMPC_GJ_U 0.01              // threshold of the pivot test of the blocked elimination (|a_kk| >= u max|a_ik|); a huge value sends every
#endif                               // edge through the out-of-line factorisation with partial pivoting (test of that fallback)
#ifndef DOMPC_GJ_ADJ
#define DOMPC_GJ_ADJ 0                // 1: inverse of the 4 x 4 pivot block from its adjugate instead of LU in uniform arithmetic + two triangular solves (measured: +-0)
#endif
#ifndef DOMPC_GJ_LTEST
#define DOMPC_GJ_LTEST 1            // 1: the threshold test of the 4 x 4 pivot blocks on the multipliers l_ik = a_ik / a_kk (|l_ik| <= 1 / u) instead of on the
#endif                              // column entries before the division: 10 instead of 18 uniform instructions per step, same decisions (+0.3 %)
#if DOMPC_GJ_LTEST && DOMPC_GJ_ADJ
#error "DOMPC_GJ_LTEST belongs to the LU variant of the pivot block"
#endif
#ifndef DOMPC_GJ_SB
#define DOMPC_GJ_SB 0               // 1: scheduling barriers at the step boundaries of the blocked elimination (measurement aid)
#endif
#if DOMPC_GJ_SB
#define GJ_SB() __builtin_amdgcn_sched_barrier(0)
#else
#define GJ_SB()
#endif
constexpr int GJ_R = DEG * NX, GJ_RP = ((GJ_R + 3) / 4) * 4, GJ_NRHS = NA + 1;
constexpr int GJ_NC = GJ_RP + GJ_NRHS + GJ_R;                      // columns: [G_cc padded | G_y r | I]
constexpr bool MFMA_GJ = (NI == 1) && (DEG >= 1) && !DENSE_EDGE && (GJ_RP <= 32) && (GJ_NC <= 64) && (DOMPC_MFMA_GJ != 0);
constexpr int GJ_MT = (GJ_RP + 15) / 16, GJ_NT = (GJ_NC + 15) / 16;
static_assert(!MFMA_GJ || GJ_RP * 4 + 256 <= EL_T1 - EL_MX, "the panel buffer and the dual-residual row share the W | w0 region of the edge working set");

// Register budget: the function is called per edge from the sweep; it must stay within the ~148 caller-saved VGPRs (every
// other register it touches costs a scratch round trip per call).  When the padded block has 16 + 4 rows (industrial_poly)
// the four rows of the second tile row are PACKED into one accumulator tile - register ni of tile X holds rows 16..19 of
// tile column ni; the MFMA that updates it gets an A operand that is zero outside rows 4 ni .. 4 ni + 3 - instead of four
// tiles with one live register each (8 instead of 32 VGPRs).
constexpr bool GJ_PACK = (GJ_MT == 2) && (GJ_RP == 20) && (GJ_NT <= 4);
constexpr int GJ_MTF = GJ_PACK ? 1 : GJ_MT;                          // full tile rows

// column descriptor of tile column ni for this lane: kind 0: G_cc (slot sl, state b), 1: y column b, 2: residual,
// 3: unit column b, 4: padding
struct GjCol { int kind, sl, b; };
__device__ inline GjCol gj_col(int ni, int lc) {
  constexpr int R = GJ_R, RP = GJ_RP, NRHS = GJ_NRHS;
  const int col = 16 * ni + lc;
  GjCol c{4, 0, 0};
  if (col < R) { c.kind = 0; c.sl = col / NX; c.b = col - c.sl * NX; }
  else if (col < RP) { c.kind = 4; c.b = col; }
  else if (col < RP + NA) { c.kind = 1; c.b = col - RP; }
  else if (col == RP + NA) { c.kind = 2; }
  else if (col < RP + NRHS + R) { c.kind = 3; c.b = col - (RP + NRHS); }
  return c;
}
// element (row, column of tile column ni) of [G_cc | G_y r | I] from the image (optimizer.py:951-963, see build_cols below)
__device__ inline double gj_element(const ldsd* mol, const ldsd* Ld, int row, int ni, int lc) {
  constexpr int R = GJ_R;
  constexpr int DG = DEG > 0 ? DEG : 1;
  const GjCol c = gj_col(ni, lc);
  const bool real = row < R;
  const int rowc = real ? row : 0;
  const int jj = rowc / NX, a = rowc - jj * NX;
  const int jcol = (c.kind == 0 || c.kind == 1) ? c.b : 0;
  const double jv = mol[(unsigned)(MO_PT + NX) + (unsigned)(jj * PT_STRIDE + a * NA + jcol)];
  const bool useJ = (c.kind == 0) ? (c.sl == jj) : (c.kind == 1 && c.b >= NX);
  double v = useJ ? jv : 0.0;
  if (c.kind == 0) {
    // C[sl + 1][jj + 1] by selects over opaque values (no constant-table load)
    double cc = 0.0;
#pragma unroll
    for (int s1 = 1; s1 <= DEG; ++s1)
#pragma unroll
      for (int j1 = 1; j1 <= DEG; ++j1) {
        double t = DOMPC_C[s1 * (DEG + 1) + j1];
        asm("" : "+v"(t));
        cc = (c.sl + 1 == s1 && jj + 1 == j1) ? t : cc;
      }
    v -= (a == c.b) ? cc : 0.0;
  }
  if (c.kind == 1) v -= (a == c.b) ? tab_sel(DOMPC_C, jj + 1, DEG > 0 ? 1 : 0, DG) : 0.0;      // C[0][jj + 1]
  if (c.kind == 2) v = Ld[EL_T1 + rowc];
  if (c.kind == 3) v = (c.b == row) ? 1.0 : 0.0;
  if (c.kind == 4) v = 0.0;
  if (!real) v = (16 * ni + lc == row) ? 1.0 : 0.0;         // padding rows: unit diagonal
  return v;
}

// Table-driven tile build.  Which entry of the image (or of the residual rows) and which constant make up element (row, column)
// of [G_cc | G_y r | I] depends on the lane and on the tile register, not on the edge: gj_element() spends ~10 vector instructions per
// element on that index arithmetic, 20 elements per lane and edge.  Once per sweep and wavefront the LDS byte offset (relative to
// the wavefront's region) of every element is written into a table behind the sweep's working set (16 bits per element and lane;
// the region belongs to the staging buffers of the Riccati passes outside the sweep): an image entry, a residual row, or a
// constant of a small pool (0, 1, -C[s][j]).  Only the diagonal of G_cc is an image entry MINUS a coefficient - those elements live
// in the tile registers whose rows and columns overlap (T[mi][mi][.], the packed register of tile column 1); their table entries
// carry the index of -C[j][j] in the three low bits (offsets are multiples of 8).  The build is then one 16-bit and one 64-bit LDS
// read per element.  Same values as gj_element() up to the sign of a zero.
#ifndef DOMPC_GJ_TABLE
#define DOMPC_GJ_TABLE 1
#endif
#ifndef DOMPC_GJ_TABLE_CHECK
#define DOMPC_GJ_TABLE_CHECK 0        // 1: build every tile both ways and trap on a difference (GPU check of the table)
#endif
constexpr int GJ_NEL = GJ_MTF * 4 * GJ_NT + (GJ_PACK ? GJ_NT : 0);      // tile registers of a lane
constexpr int GJ_NPOOL = 2 + (DEG + 1) * DEG;                           // 0, 1, -C[s][j] (s = 0..DEG, j = 1..DEG)
constexpr int GJ_TAB = ((EL_MOC + MOC_STAGE + 1) / 2) * 2;
constexpr int GJ_POOL = GJ_TAB + (GJ_NEL * 64 * 2 + 7) / 8;
constexpr int GJ_DPOOL = GJ_POOL + GJ_NPOOL;                            // 0, -C[1][1], ..., -C[DEG][DEG]
constexpr bool GJ_TABLE = MFMA_GJ && (DOMPC_GJ_TABLE != 0) && (GJ_DPOOL + DEG + 1 <= EL_SIZE) && (EL_SIZE <= 2048) && (DEG <= 7);
typedef __attribute__((address_space(3))) unsigned short ldsu16;
typedef __attribute__((address_space(3))) char ldsc;
// table entry of element (row, column lc of tile column ni): mirrors gj_element()
__device__ inline unsigned gj_entry(int row, int ni, int lc) {
  constexpr int R = GJ_R;
  const GjCol c = gj_col(ni, lc);
  unsigned off = GJ_POOL, ci = 0, dg = 0;         // (pool entry 0 is 0.0)
  if (row >= R) {
    ci = (16 * ni + lc == row) ? 1u : 0u;
  } else {
    const int jj = row / NX, a = row - jj * NX;
    const unsigned jo = (unsigned)(EL_MOS + MO_PT + NX) + (unsigned)(jj * PT_STRIDE + a * NA);
    if (c.kind == 0) {
      if (c.sl == jj) off = jo + (unsigned)c.b;
      if (a == c.b) {
        if (c.sl == jj) dg = (unsigned)(jj + 1);          // diagonal of G_cc: image entry - C[jj + 1][jj + 1]
        else ci = 2u + (unsigned)((c.sl + 1) * DEG + jj);
      }
    } else if (c.kind == 1) {
      if (c.b >= NX) off = jo + (unsigned)c.b;
      else if (a == c.b) ci = 2u + (unsigned)jj;
    } else if (c.kind == 2) {
      off = (unsigned)(EL_T1 + row);
    } else if (c.kind == 3) {
      ci = (c.b == row) ? 1u : 0u;
    }
  }
  if (ci) off = GJ_POOL + ci;                     // (never together with an image entry)
  return (off << 3) | dg;
}
// once per sweep and wavefront (all 64 lanes of the wavefront that owns Ld)
__device__ inline void gj_table_init(ldsd* Ld, int lane) {
  if constexpr (GJ_TABLE) {
    ldsu16* tab = (ldsu16*)(Ld + GJ_TAB);
    const int lr = lane >> 4, lc = lane & 15;
    int el = 0;
#pragma unroll
    for (int mi = 0; mi < GJ_MTF; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ni = 0; ni < GJ_NT; ++ni, ++el)
          tab[el * 64 + lane] = (unsigned short)((16 * mi + 4 * r >= GJ_RP) ? (unsigned)(GJ_POOL << 3) : gj_entry(16 * mi + 4 * r + lr, ni, lc));
    if constexpr (GJ_PACK) {
#pragma unroll
      for (int ni = 0; ni < GJ_NT; ++ni, ++el) tab[el * 64 + lane] = (unsigned short)gj_entry(16 + lr, ni, lc);
    }
    if (lane < GJ_NPOOL) {
      double v = (lane == 1) ? 1.0 : 0.0;
      if (lane >= 2) v = -DOMPC_C[((lane - 2) / DEG) * (DEG + 1) + (lane - 2) % DEG + 1];
      Ld[GJ_POOL + lane] = v;
    }
    if (lane <= DEG) Ld[GJ_DPOOL + lane] = (lane == 0) ? 0.0 : -DOMPC_C[lane * (DEG + 1) + lane];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <class DUAL>
__device__ inline int edge_factor_mfma(const Prob& Q, int e, int lane, ldsd* Ld, DUAL&& dual_from) {
  constexpr int R = GJ_R, RP = GJ_RP, MT = GJ_MTF > 0 ? GJ_MTF : 1, NT = GJ_NT > 0 ? GJ_NT : 1;      // (at least one tile: the function is compiled for every model)
  constexpr double GJ_U = DOMPC_GJ_U;
  const ldsd* mol = Ld + EL_MOS;                  // dense image of the model-output record
  ldsd* pan = Ld + EL_MX;                         // panel columns of the current step, RP x 4 row-major
#if DOMPC_PROFILE
  long long pc0_ = clock64();
#define GJ_PH(i) if (threadIdx.x == 0) { const long long pc1_ = clock64(); lds_prof[i] += pc1_ - pc0_; pc0_ = pc1_; }
#else
#define GJ_PH(i)
#endif
  const int lr = lane >> 4, lc = lane & 15;
#if DOMPC_GJ_PRIO
  __builtin_amdgcn_s_setprio(DOMPC_GJ_PRIO);
#endif
  d4 T[MT][NT];
  d4 X = {0.0, 0.0, 0.0, 0.0};                    // GJ_PACK: register ni = rows 16..19 of tile column ni
  // ---- tiles of [G_cc | G_y r | I]
  if constexpr (GJ_TABLE) {
    const ldsu16* tab = (const ldsu16*)(Ld + GJ_TAB) + lane;
    const ldsc* Lb = (const ldsc*)Ld;
    auto elem = [&](int el, bool diag) {            // (diag: compile-time - the register can hold diagonal entries of G_cc)
      const unsigned w = tab[el * 64];
      if (!diag) return (double)*(const ldsd*)(Lb + w);
      return (double)*(const ldsd*)(Lb + (w & 0xfff8u)) + (double)Ld[GJ_DPOOL + (w & 7u)];
    };
    int el = 0;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni, ++el) T[mi][ni][r] = elem(el, ni == mi);
    if constexpr (GJ_PACK) {
#pragma unroll
      for (int ni = 0; ni < NT; ++ni, ++el) X[ni] = elem(el, ni == 1);
    }
#if DOMPC_GJ_TABLE_CHECK
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          if (T[mi][ni][r] != ((16 * mi + 4 * r >= RP) ? 0.0 : gj_element(mol, Ld, 16 * mi + 4 * r + lr, ni, lc))) __builtin_trap();
    if constexpr (GJ_PACK) {
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
        if (X[ni] != gj_element(mol, Ld, 16 + lr, ni, lc)) __builtin_trap();
    }
#endif
  } else {
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
        T[mi][ni][r] = (16 * mi + 4 * r >= RP) ? 0.0 : gj_element(mol, Ld, 16 * mi + 4 * r + lr, ni, lc);
  if constexpr (GJ_PACK) {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) X[ni] = gj_element(mol, Ld, 16 + lr, ni, lc);
  }
  }
  GJ_PH(25)
  {
    // ---- dual-residual pieces: lambda' [G_cc | G_y] on the matrix cores.  A operand: the multipliers of the collocation rows
    // in row 0 of a 16 x 4 block per k-block; B operand: the tile registers themselves (register r of tile row mi = rows
    // 16 mi + 4 r ...).  Row 0 of the result tiles goes through LDS to the lanes that own the columns (dual_from).
    constexpr int NDT = (RP + NA + 15) / 16 < NT ? (RP + NA + 15) / 16 : NT;
    ldsd* du = Ld + EL_MX + 4 * RP;                 // (behind the panel buffer; the W | w0 region is written after the last step)
#if DOMPC_DUAL_VALU
    // on the vector ALU: this lane's rows of its columns (4 per full tile row + 1 packed) times their multipliers; the four lane
    // groups of a column leave their partial sums in four rows of the buffer, the reader adds them (an MFMA with the multipliers
    // in one row of the A operand does the same at 1/16 of its throughput: 15 instructions of 64 cycles)
    {
      double lamr[MT][4], lamx = 0.0;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * mi + 4 * r + lr;
          lamr[mi][r] = (row < R) ? (double)Ld[EL_T0 + row] : 0.0;
        }
      if constexpr (GJ_PACK) lamx = (16 + lr < R) ? (double)Ld[EL_T0 + 16 + lr] : 0.0;
#pragma unroll
      for (int ni = 0; ni < NDT; ++ni) {
        double t = 0.0;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) t = fma(lamr[mi][r], T[mi][ni][r], t);
        if constexpr (GJ_PACK) t = fma(lamx, X[ni], t);
        du[64 * lr + 16 * ni + lc] = t;
      }
    }
#else
    d4 acc[NDT];
#pragma unroll
    for (int ni = 0; ni < NDT; ++ni) acc[ni] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kb = 0; kb < RP / 4; ++kb) {
      const int row = 4 * kb + lr;
      const double lam = Ld[EL_T0 + (row < R ? row : 0)];
      const double a = (lc == 0 && row < R) ? lam : 0.0;
#pragma unroll
      for (int ni = 0; ni < NDT; ++ni) {
        const double b = (GJ_PACK && kb >= 4) ? X[ni] : T[(GJ_PACK && kb >= 4) ? 0 : kb / 4][ni][kb % 4];
        acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[ni], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ni = 0; ni < NDT; ++ni) {              // (same buffer layout as the vector-ALU variant: row 0 holds the sums)
      du[64 * lr + 16 * ni + lc] = (lr == 0) ? acc[ni][0] : 0.0;
    }
#endif
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    dual_from((const ldsd*)du);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  GJ_PH(24)
  double viol = -1.0, pmin = 1.0;
  // ---- RP / 4 steps of four pivots
#pragma unroll
  for (int p = 0; p < RP / 4; ++p) {
    GJ_SB();
    const int mip = p / 4, rp = p % 4, nip = p / 4, c0 = 4 * (p % 4);
    const bool prow_x = GJ_PACK && mip == 1;       // (the panel rows live in the packed tile)
    // panel columns -> LDS
    if (lc >= c0 && lc < c0 + 4) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * mi + 4 * r < RP) pan[(16 * mi + 4 * r + lr) * 4 + (lc - c0)] = T[mi][nip][r];
      if constexpr (GJ_PACK) pan[(16 + lr) * 4 + (lc - c0)] = X[nip];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // the panel rows as they are now: B operands of the update
    double Rb[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) Rb[ni] = prow_x ? X[ni] : T[prow_x ? 0 : mip][ni][rp];
    double a_[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) a_[i][j] = pan[(4 * p + i) * 4 + j];
    double x_[4];
#if DOMPC_GJ_ADJ
    // column lr of P^-1 from the adjugate (2 x 2 minors of the row pairs (0,1) and (2,3), Laplace expansion): a dependent chain of
    // ~12 instructions instead of ~43 through the LU factors and the two triangular solves - a dependent FP64 instruction costs
    // ~16 cycles here, and this chain sits in front of the matrix-core instructions of every step.  Accepted if the determinant
    // lost less than four digits to cancellation (|det| >= 1e-4 sum |terms|); otherwise the caller repeats the factorisation with
    // partial pivoting like after a failed threshold test of the LU variant.
    {
      const int rho = 4 * p + (lr ^ 1);                         // column j of the adjugate is built from row j ^ 1 and the minors of the OTHER row pair
      const double r0 = pan[rho * 4 + 0], r1 = pan[rho * 4 + 1], r2 = pan[rho * 4 + 2], r3 = pan[rho * 4 + 3];
      double sm[6], cm[6];
      constexpr int MA[6] = {0, 0, 0, 1, 1, 2}, MB[6] = {1, 2, 3, 2, 3, 3};      // column pairs of the minors
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        sm[k] = fma(a_[0][MA[k]], a_[1][MB[k]], -(a_[1][MA[k]] * a_[0][MB[k]]));
        cm[k] = fma(a_[2][MA[k]], a_[3][MB[k]], -(a_[3][MA[k]] * a_[2][MB[k]]));
      }
      const double t0 = sm[0] * cm[5], t1 = sm[1] * cm[4], t2 = sm[2] * cm[3], t3 = sm[3] * cm[2], t4 = sm[4] * cm[1], t5 = sm[5] * cm[0];
      const double det = ((t0 - t1) + (t2 + t3)) + (t5 - t4);
      const double mag = ((fabs(t0) + fabs(t1)) + (fabs(t2) + fabs(t3))) + (fabs(t5) + fabs(t4));
      viol = fmax(viol, fma(1e-4, mag, -fabs(det)));            // > 0: cancellation
      pmin = fmin(pmin, fabs(det));
      double m_[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) m_[k] = (lr < 2) ? cm[k] : sm[k];
      const double idet = fast_rcp((fabs(det) > 1e-300) ? det : 1.0);
      const double sg = (lr & 1) ? -idet : idet;
      x_[0] = sg * fma(r1, m_[5], fma(-r2, m_[4], r3 * m_[3]));
      x_[1] = sg * fma(-r0, m_[5], fma(r2, m_[2], -(r3 * m_[1])));
      x_[2] = sg * fma(r0, m_[4], fma(-r1, m_[2], r3 * m_[0]));
      x_[3] = sg * fma(-r0, m_[3], fma(r1, m_[1], -(r2 * m_[0])));
    }
#else
    // P, LU in uniform arithmetic
    double iu[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#if !DOMPC_GJ_LTEST
      double m = 0.0;
#pragma unroll
      for (int i = k + 1; i < 4; ++i) m = fmax(m, fabs(a_[i][k]));
      viol = fmax(viol, fma(GJ_U, m, -fabs(a_[k][k])));      // > 0: |a_kk| < GJ_U max|a_ik|
#endif
      pmin = fmin(pmin, fabs(a_[k][k]));
      iu[k] = fast_rcp(a_[k][k]);
#pragma unroll
      for (int i = k + 1; i < 4; ++i) {
        a_[i][k] *= iu[k];
#if DOMPC_GJ_LTEST
        viol = fmax(viol, fabs(a_[i][k]));                  // the same test on the multipliers: |l_ik| <= 1 / GJ_U
#endif
#pragma unroll
        for (int j = k + 1; j < 4; ++j) a_[i][j] = fma(-a_[i][k], a_[k][j], a_[i][j]);
      }
    }
    // column k = lr of P^-1:  L y = e_k, U x = y
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double t = (lr == i) ? 1.0 : 0.0;
#pragma unroll
      for (int j = 0; j < i; ++j) t = fma(-a_[i][j], x_[j], t);
      x_[i] = t;
    }
#pragma unroll
    for (int i = 3; i >= 0; --i) {
      double t = x_[i];
#pragma unroll
      for (int j = i + 1; j < 4; ++j) t = fma(-a_[i][j], x_[j], t);
      x_[i] = t * iu[i];
    }
#endif
    GJ_SB();             // (the LU factors are dead: do not hoist the loads below above them)
    // this lane's entries of -(C~ P^-1): row lc of every full tile row (packed rows: row 16 + (lc & 3)), column lr
    auto cprime = [&](int row) {
      const int rowc = row < RP ? row : 0;
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double cj = pan[rowc * 4 + j] - ((row == 4 * p + j) ? 1.0 : 0.0);
        t = fma(cj, x_[j], t);
      }
      return (row < RP) ? -t : 0.0;
    };
    double cp[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) cp[mi] = cprime(16 * mi + lc);
    double cpx = 0.0;
    if constexpr (GJ_PACK) cpx = cprime(16 + (lc & 3));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // rank-4 update of the tiles that still hold columns to the right of the panel
    // (the tile column that holds the NEXT panel first: its columns are needed at the top of the next step, whose LU
    //  arithmetic then runs under the remaining matrix-core instructions)
#pragma unroll
    for (int o = 0; o < NT; ++o) {
      const int nxt = (p + 1) / 4 < NT ? (p + 1) / 4 : 0;
      const int ni = (o == 0) ? nxt : (o <= nxt ? o - 1 : o);
      if (16 * (ni + 1) <= 4 * (p + 1)) continue;
      // a tile column that holds only unit columns e_b (and padding) with b >= 4 (p + 1): their entries in the panel rows are still
      // zero - the update would add nothing (industrial_poly: tile column 3 during the first three steps, 6 of 36 MFMAs)
      if (DOMPC_GJ_SKIP && 16 * ni >= RP + GJ_NRHS && 16 * ni - (RP + GJ_NRHS) >= 4 * (p + 1)) continue;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) T[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(cp[mi], Rb[ni], T[mi][ni], 0, 0, 0);
      if constexpr (GJ_PACK) X = __builtin_amdgcn_mfma_f64_16x16x4f64(((lc >> 2) == ni) ? cpx : 0.0, Rb[ni], X, 0, 0, 0);
    }
  }
  GJ_PH(26)
#if DOMPC_GJ_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
#if DOMPC_GJ_LTEST
  if (!(viol <= 1.0 / GJ_U && pmin > 1e-300)) return 1;  // (NaN-safe: a failed test or a vanishing pivot)
#else
  if (!(viol <= 0.0 && pmin > 1e-300)) return 1;        // (NaN-safe: a failed test or a vanishing pivot)
#endif
  // ---- W | w0 (collocation rows) -> LDS, G_cc^-1 -> forward record
  auto put = [&](int row, int ni, double v) {
    const GjCol c = gj_col(ni, lc);
    if (row < R) {
      if (c.kind == 1 || c.kind == 2) Ld[EL_MX + row * MX_LD + MX_W + (c.kind == 2 ? NA : c.b)] = -v;
      if (c.kind == 3) Q.EW(e, EW_LU + row * LU_N + c.b) = v;
    }
  };
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
    if (16 * ni + 15 < RP) continue;             // (columns of the eliminated block)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * mi + 4 * r < RP) put(16 * mi + 4 * r + lr, ni, T[mi][ni][r]);
    if constexpr (GJ_PACK) put(16 + lr, ni, X[ni]);
  }
  GJ_PH(27)
#undef GJ_PH
  return 0;
}
#else
constexpr bool MFMA_GJ = false;
DOMPC_DEV inline void gj_table_init(ldsd*, int) {}
#endif

// ================================================================================================
// Single finite element: factorisation part of an edge - columns of [G_cc | G_y r | I] in registers, dual-residual pieces,
// register-resident Gauss-Jordan (with its pivoting fallback), W | w0 into LDS, G_cc^-1 to the forward record.
// Its own function on the device (phase_edge_factor, noinline): the elimination is the most register- and schedule-
// sensitive code of the kernel (adding four live values in front of it cost 15 %, removing its never-executed fallback
// made it 7x slower when it shared a function with the assembly and condensing code); on its own it gets the whole
// register file and a schedule that does not depend on what surrounds the call.
// In: Ld[EL_T1] residual rows, Ld[EL_T0] multipliers of the edge's rows, the staged model-output record; this lane's
// per-variable data (vx: its extended column, ex / nu_a: end-point column on the first NX lanes).
constexpr int EF_R = DEG * NX, EF_NCX = 2 * EF_R + NA + 1, EF_CPX = (EF_NCX + GS_C - 1) / GS_C;
// MODE 0: everything with the register-resident elimination (host emulation, models outside the matrix-core variant);
// MODE 1 (device, MFMA_GJ): dual-residual pieces + blocked elimination on the matrix cores, returns 2 if its threshold test
//        fails; MODE 2 (device, MFMA_GJ): the repeat in that case - columns, elimination with partial pivoting, outputs.
template <int MODE>
DOMPC_DEV inline int edge_factor_body(const Prob& Q, int e, double mu, int lane, int GS, ldsd* Ld, const double (&vx)[EF_CPX][5],
                                      const double (&ex)[5], double nu_a) {
  const KArgs& A = *Q.A;
  const int woff = A.edge_w_off[e];
  const double* nu_e = Q.lam + A.edge_row0[e] + NW;
  const double* mo = Q.MO(e);
  (void)mo;
  const ldsd* mol = Ld + EL_MOS;          // (dense image of the compact record, see mo_expand)
#define MOV(i) (MO_COMPACT ? (double)mol[(i)] : mo[(i)])
  const bool act = true;
  int fail = 0;
  (void)act;
#if DOMPC_PROFILE && !defined(DOMPC_HOST_EMU)
  const long long pc_ef0 = clock64();
#endif
  constexpr int R = DEG * NX, RA = R > 0 ? R : 1;
  constexpr int NRHS = NA + 1;
  constexpr int NCX = 2 * R + NRHS;                      // extended columns: G_cc | G_y r | I
  constexpr int CPX = (NCX + GS_C - 1) / GS_C;
  constexpr double GJ_U = DOMPC_GJ_U;        // (threshold of the natural pivot order, as in the blocked variant)
  double bc[CPX][RA];
  // column cx of the collocation rows (row r = (jj, a): point j = jj + 1, state a), straight from the model-output
  // record (optimizer.py:951-963):  G_cc (slot sl, state b): [sl == jj] J_jj[a][b] - [a == b] C[sl+1][j];
  // G_y: x_n columns -[a == yb] C[0][j], u_n columns J_jj[a][yb];  r: the residuals (staged in LDS by the lanes
  // that computed them);  I.  One unconditional load per entry (clamped address) + selects: no divergent branches.
  // (all global loads first, in one batch - fetch_cols(), called before anything of this edge is computed: loads
  //  issued between dependent selects / branches are waited for one by one; the first version of this assembly
  //  spent 40 serialized memory round trips per edge that way)
  double cd[CPX][DEG > 0 ? DEG : 1];
  unsigned jcol_[CPX];
  auto fetch_cols = [&]() {
#pragma unroll
    for (int q = 0; q < CPX; ++q) {
      const unsigned cx = (unsigned)lane + (unsigned)q * (unsigned)GS;
      const bool isG = cx < (unsigned)R, isY = cx >= (unsigned)R && cx < (unsigned)(R + NA);
      const unsigned jcol = isG ? cx % (unsigned)NX : (isY ? cx - (unsigned)R : 0u);   // column of the point Jacobian this lane reads
      const unsigned sl1 = isG ? cx / (unsigned)NX + 1u : 0u;
      jcol_[q] = jcol;
#if !defined(DOMPC_HOST_EMU)
      // the diagonal collocation coefficient of this column by selects over opaque values: an indexed read of the constant
      // table would be the only global load of this function - a full memory round trip in front of the elimination
#pragma unroll
      for (int jj = 0; jj < DEG; ++jj) {
        double v = DOMPC_C[jj + 1];
        asm("" : "+v"(v));
#pragma unroll
        for (int s1 = 1; s1 <= DEG; ++s1) {
          double t = DOMPC_C[s1 * (DEG + 1) + jj + 1];
          asm("" : "+v"(t));
          v = (sl1 == (unsigned)s1) ? t : v;
        }
        cd[q][jj] = v;
      }
#else
#pragma unroll
      for (int jj = 0; jj < DEG; ++jj) cd[q][jj] = DOMPC_C[sl1 * (unsigned)(DEG + 1) + (unsigned)(jj + 1)];
#endif
    }
  };
  auto build_cols = [&]() {
#pragma unroll
    for (int q = 0; q < CPX; ++q) {
      const int cx = lane + q * GS;
      const bool isG = cx < R, isY = cx >= R && cx < R + NA, isR = cx == R + NA;
      const int sl = isG ? cx / NX : -1, b = isG ? cx % NX : -1, yb = isY ? cx - R : -1;
      const int unit_row = cx - (R + NRHS);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int jj = r / NX, a = r % NX;
        const bool useJ = isG ? (sl == jj) : (isY && yb >= NX);
        const double jv_ = MOV((unsigned)(MO_PT + (r / NX) * PT_STRIDE + NX + (r % NX) * NA) + jcol_[q]);   // (entry of the point Jacobian, read where it is used: no second 20-entry array alive next to the column)
        double v = useJ ? jv_ : 0.0;
        v -= (a == b) ? cd[q][jj] : 0.0;
        v -= (a == yb) ? DOMPC_C[0 * (DEG + 1) + (jj + 1)] : 0.0;
        v = (unit_row == r) ? 1.0 : v;
        bc[q][r] = v;
      }
      if (isR) {
#pragma unroll
        for (int r = 0; r < R; ++r) bc[q][r] = Ld[EL_T1 + r];
      }
    }
  };
  auto eliminate = [&](bool pivoting) -> int {           // returns 1: threshold test failed / singular block
    int badl = 0;
#pragma unroll
    for (int kk = 0; kk < R; ++kk) {
      const int qk = kk / GS_C, lk = kk % GS_C;          // column kk lives in slot qk of lane lk
      if (pivoting) {
        int pr = kk;
        double best = fabs(bc[qk][kk]);
#pragma unroll
        for (int r = kk + 1; r < R; ++r) {
          const double a = fabs(bc[qk][r]);
          if (a > best) { best = a; pr = r; }
        }
        if (lane == lk && !(best > 1e-300)) badl = 1;
#ifndef DOMPC_HOST_EMU
        pr = __builtin_amdgcn_readlane(pr, lk);
#endif
#pragma unroll
        for (int q = 0; q < CPX; ++q) {                  // rows kk <-> pr (the appended identity is permuted along)
          const double t = bc[q][kk];
          double nk = t;
#pragma unroll
          for (int r = kk + 1; r < R; ++r) {
            const bool hit = (r == pr);
            nk = hit ? bc[q][r] : nk;
            bc[q][r] = hit ? t : bc[q][r];
          }
          bc[q][kk] = nk;
        }
      } else {
        double m = 0.0;
#pragma unroll
        for (int r = kk + 1; r < R; ++r) m = fmax(m, fabs(bc[qk][r]));
        const double akk = fabs(bc[qk][kk]);
        if (lane == lk && !(akk >= GJ_U * m && akk > 1e-300)) badl = 1;
      }
      double f[RA];
#pragma unroll
      for (int r = 0; r < R; ++r) f[r] = lane_bcast(bc[qk][r], lk);
      const double pinv = fast_rcp((fabs(f[kk]) > 1e-300) ? f[kk] : 1.0);
#pragma unroll
      for (int q = 0; q < CPX; ++q) {
        const double prow = bc[q][kk] * pinv;
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (r != kk) bc[q][r] = fma(-f[r], prow, bc[q][r]);
        bc[q][kk] = prow;
      }
    }
#ifndef DOMPC_HOST_EMU
    return __ballot(badl) != 0ull;
#else
    return badl;
#endif
  };
  if (MODE != 1) fetch_cols();
  // dual-residual pieces: column c of G_w / G_y times the multipliers of the edge's rows (continuity rows:
  // -D_{sl+1} on the diagonal of the G_cc columns, -D_0 for the x_n columns, +1 for the end-point columns).
  // `col_dot(q, cx)`: the collocation rows' share of column cx (MODE 0: from the column in registers; MODE 1: formed by
  // the matrix cores from the tiles, edge_factor_mfma)
  auto dual_pieces = [&](auto col_dot) {
#pragma unroll
    for (int q = 0; q < CPX; ++q) {
      const int cx = lane + q * GS;
      double t = col_dot(q, cx);
      if (cx < R) {
        t -= DOMPC_D[cx / NX + 1] * Ld[EL_T0 + R + cx % NX];   // (measured: neither a select chain nor a load of the coefficient in the first batch of the edge pays - both slow the elimination that follows by more than the round trip they save)
        const int gi = woff + cx;
        const double xv = vx[q][0], l = vx[q][1], u = vx[q][2], zl_ = vx[q][3], zu_ = vx[q][4];
        Q.gf[gi] = 0.0;
        Q.rd[gi] = t - zl_ + zu_;
        Ld[EL_RW + cx] = t + bar_grad(xv, l, u, mu, !(Q.soc & 2));
        Ld[EL_BB + cx] = bar_grad(xv, l, u, 1.0);
        Ld[EL_SG + cx] = sigma_of(xv, l, u, zl_, zu_);
      } else if (cx < R + NA) {
        const int yb = cx - R;
        if (yb < NX) t -= DOMPC_D[0] * Ld[EL_T0 + R + yb];
        Ld[EL_RY + yb] = t;          // completed in phase 7
      }
    }
    for (int a = lane; a < NX; a += GS) {               // end-point (xkf) columns
      const int col = R + a, gi = woff + col;
      const double t = Ld[EL_T0 + R + a] + (GS > 1 ? nu_a : nu_e[a]);
      double xv, l, u, zl_, zu_;
      if (GS > 1) { xv = ex[0]; l = ex[1]; u = ex[2]; zl_ = ex[3]; zu_ = ex[4]; }
      else { xv = Q.x[gi]; l = Q.lb[gi]; u = Q.ub[gi]; zl_ = Q.zl[gi]; zu_ = Q.zu[gi]; }
      Q.gf[gi] = 0.0;
      Q.rd[gi] = t - zl_ + zu_;
      Ld[EL_RW + col] = t + bar_grad(xv, l, u, mu, !(Q.soc & 2));
      Ld[EL_BB + col] = bar_grad(xv, l, u, 1.0);
      Ld[EL_SG + col] = sigma_of(xv, l, u, zl_, zu_);
    }
  };
  if (act && MODE == 0) {
    build_cols();
    dual_pieces([&](int q, int) {
      double t = 0.0;
#pragma unroll
      for (int r = 0; r < R; ++r) t += bc[q][r] * Ld[EL_T0 + r];
      return t;
    });
  }
#ifndef DOMPC_HOST_EMU
  if constexpr (MODE == 1) {
    // blocked elimination on the matrix cores (edge_factor_mfma); if its threshold test fails the caller repeats the
    // factorisation with the register-resident elimination and partial pivoting (MODE 2, its own out-of-line function:
    // this one stays within the caller-saved registers - a callee pays a scratch round trip for every other one it touches)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (edge_factor_mfma(Q, e, lane, Ld, [&](const ldsd* du) {
          dual_pieces([&](int, int cx) {
            const int c_ = cx < R ? cx : (cx < R + NA ? GJ_RP + (cx - R) : 0);
            return (double)du[c_] + (double)du[64 + c_] + (double)du[128 + c_] + (double)du[192 + c_];
          });
        })) return 2;
    {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // continuity rows of W | w0:  W_e = D_0 [I 0] + sum_s D_s W_s ,  w0_e = -r_e + sum_s D_s w0_s
      for (int it = lane; it < NX * NRHS; it += GS) {
        const int a_ = it / NRHS, c = it % NRHS;
        double t = (c == NA) ? -Ld[EL_T1 + R + a_] : ((c == a_) ? DOMPC_D[0] : 0.0);
#pragma unroll
        for (int s_ = 1; s_ <= DEG; ++s_) t += DOMPC_D[s_] * Ld[EL_MX + ((s_ - 1) * NX + a_) * MX_LD + MX_W + c];
        Ld[EL_MX + (R + a_) * MX_LD + MX_W + c] = t;
      }
      return 0;
    }
  } else if constexpr (MODE == 2) {
    build_cols();
    if (eliminate(true)) fail = 1;
  } else
#endif
  if (act) {
    if (eliminate(false)) {
      fetch_cols();
      build_cols();
      if (eliminate(true)) fail = 1;
    }
  }
  if (act) {
#pragma unroll
    for (int q = 0; q < CPX; ++q) {
      const int cx = lane + q * GS;
      if (cx >= R && cx < R + NRHS) {
        // right-hand sides: W = -G_w^-1 G_y, w0 = -G_w^-1 r_g; continuity row a = assembled entry + sum_r D_r row((r-1)NX+a)
        const int col = MX_W + (cx - R);
#pragma unroll
        for (int a_ = 0; a_ < NX; ++a_) {
          const int yb = cx - R;                       // assembled entry of the continuity row: -D_0 / the residual
          double t = (yb == NA) ? Ld[EL_T1 + R + a_] : ((yb == a_) ? -DOMPC_D[0] : 0.0);
#pragma unroll
          for (int r = 1; r <= DEG; ++r) t += DOMPC_D[r] * bc[q][(r - 1) * NX + a_];
          Ld[EL_MX + (R + a_) * MX_LD + col] = -t;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) Ld[EL_MX + r * MX_LD + col] = -bc[q][r];
      } else if (cx >= R + NRHS && cx < NCX) {
        const int col = cx - (R + NRHS);                 // column `col` of G_cc^-1 (kept for the multiplier recovery)
#pragma unroll
        for (int r = 0; r < R; ++r) Q.EW(e, EW_LU + r * LU_N + col) = bc[q][r];
      }
    }
  }
#undef MOV
  return fail;
}

#ifndef DOMPC_HOST_EMU
__device__ inline KArgs kernel_args(const void* kp);
#define DOMPC_EF_ARGS const void* kp, int slot, int e, int soc, double sf, double mu, double v0, double v1, double v2, double v3, double v4, \
                      double x0, double x1, double x2, double x3, double x4, double nu_a
#define DOMPC_EF_BODY(MODE_)                                                                                     \
  const KArgs A = kernel_args(kp);                                                                               \
  Prob Q = make_prob(A, __builtin_amdgcn_readfirstlane(slot), nullptr);                                          \
  Q.sf = ufl(sf);                                                                                                \
  Q.soc = __builtin_amdgcn_readfirstlane(soc);                                                                   \
  prob_bounds(Q);                                                                                                \
  const int lane = (int)(threadIdx.x & 63u);                                                                     \
  ldsd* Ld = (ldsd*)lds_pool + (int64_t)(threadIdx.x >> 6) * EL_SIZE;                                            \
  const double vx[EF_CPX][5] = {{v0, v1, v2, v3, v4}};                                                           \
  const double ex[5] = {x0, x1, x2, x3, x4};                                                                     \
  return edge_factor_body<MODE_>(Q, __builtin_amdgcn_readfirstlane(e), ufl(mu), lane, 64, Ld, vx, ex, nu_a);
__device__ __attribute__((noinline)) int phase_edge_factor(DOMPC_EF_ARGS) { DOMPC_EF_BODY(MFMA_GJ ? 1 : 0) }
__device__ __attribute__((noinline)) int phase_edge_factor_pivot(DOMPC_EF_ARGS) { DOMPC_EF_BODY(MFMA_GJ ? 2 : 0) }
#undef DOMPC_EF_BODY
#undef DOMPC_EF_ARGS
#endif
DOMPC_DEV inline int run_edge_factor(const Thr& T, const Prob& Q, int e, double mu, int lane, int GS, ldsd* Ld,
                                     const double (&vx)[EF_CPX][5], const double (&ex)[5], double nu_a) {
#ifndef DOMPC_HOST_EMU
  if constexpr (EF_CPX == 1) {
    (void)lane; (void)GS; (void)Ld;
#ifndef DOMPC_EF_INLINE
#define DOMPC_EF_INLINE 0          // 1: the matrix-core factorisation inside the sweep function (no call, no callee-saved registers to save per edge)
#endif
    int rc;
    if constexpr (MFMA_GJ && DOMPC_EF_INLINE)
      rc = edge_factor_body<1>(Q, e, mu, lane, GS, Ld, vx, ex, nu_a);
    else
      rc = phase_edge_factor(T.kp, Q.slot, e, Q.soc, Q.sf, mu, vx[0][0], vx[0][1], vx[0][2], vx[0][3], vx[0][4],
                             ex[0], ex[1], ex[2], ex[3], ex[4], nu_a);
    if (MFMA_GJ && __builtin_amdgcn_readfirstlane(rc) == 2)          // (threshold test of the blocked elimination failed: rare)
      rc = phase_edge_factor_pivot(T.kp, Q.slot, e, Q.soc, Q.sf, mu, vx[0][0], vx[0][1], vx[0][2], vx[0][3], vx[0][4],
                                   ex[0], ex[1], ex[2], ex[3], ex[4], nu_a);
    return rc;
  } else {
    return edge_factor_body<0>(Q, e, mu, lane, GS, Ld, vx, ex, nu_a);
  }
#else
  (void)T;
  return edge_factor_body<0>(Q, e, mu, lane, GS, Ld, vx, ex, nu_a);
#endif
}

#ifndef DOMPC_HOST_EMU
// request the copy of the model-output record of edge e into the wavefront's staging area (LDS-DMA: global_load_lds_dwordx4,
// 64 lanes x 16 B per instruction, no staging registers; completion is awaited with s_waitcnt vmcnt).  The last piece may run
// past the end of the record into the next one / the slack behind the array (ws_layout) - never used.
__device__ inline void stage_mo(const Prob& Q, int e, int lane, ldsd* Ld) {
  const double* src = Q.MO(e);
#pragma unroll
  for (int q = 0; q < MOC_STAGE / 128; ++q)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 128 * q + 2 * lane),
                                     (__attribute__((address_space(3))) void*)(Ld + EL_MOC + 128 * q), 16, 0, 0);
}
#endif

// `staged_e` (device, single finite element): the edge whose model-output record is in (or on its way into) the staging area
// of this wavefront; the function requests the record of `e_next` as soon as it has read the last entry of its own.
DOMPC_PHASE int eval_edge_coop(const Thr& T, const Prob& Q, int e, int e_next, double mu, int lane, int GS, ldsd* Ld, int& staged_e, const MocMap& mm) {
  const KArgs& A = *Q.A;
  const bool act = e >= 0;
  const int ee = act ? e : 0;
#ifndef DOMPC_EDGE_PACK
#define DOMPC_EDGE_PACK 1           // the indices of an edge from its packed record (KArgs::edge_pack); 0: from the separate tables
#endif
#if DOMPC_EDGE_PACK
  const auto* ep = A.edge_pack + ee * EP_N;             // (the edge's indices side by side: one scalar load, dompc_kargs.h)
  const int n = ep[EP_PARENT], cn = ep[EP_CHILD], k = ep[EP_LEVEL];
  const double* xn = Q.x + ep[EP_XOFF_PARENT];
  const double* un = Q.x + ep[EP_UOFF_PARENT];
  const double* xc = Q.x + ep[EP_XOFF_CHILD];
  const int woff = ep[EP_WOFF];
  const int eps_off_n = ep[EP_EPSOFF_PARENT];
  const double* pp = Q.P + A.p_off_p + ep[EP_PIDX] * NP;
  const int row0 = ep[EP_ROW0];
  const double om = __builtin_bit_cast(double, ((unsigned long long)(unsigned)ep[EP_OMEGA_HI] << 32) | (unsigned long long)(unsigned)ep[EP_OMEGA_LO]) * Q.sf;
#else
  const int n = A.edge_parent[ee], cn = A.edge_child[ee], k = A.edge_level[ee];
  const double* xn = Q.x + A.node_x_off[n];
  const double* un = Q.x + A.node_u_off[n];
  const double* xc = Q.x + A.node_x_off[cn];
  const int woff = A.edge_w_off[ee];
  const int eps_off_n = NSE > 0 ? A.node_eps_off[n] : 0;
  const double* pp = Q.P + A.p_off_p + A.edge_pidx[ee] * NP;
  const int row0 = A.edge_row0[ee];
  const double om = A.edge_omega[ee] * Q.sf;
#endif
  (void)eps_off_n;
  const double* w = Q.x + woff;
  const double* tvp = Q.P + A.p_off_tvp + k * NTVP;
  const double omh = (Q.soc & 2) ? 0.0 : om;          // weight of the objective HESSIANS (Prob::soc bit 1)
  const double* lam_e = Q.lam + row0;
  const double* nu_e = Q.lam + row0 + NW;
  const double* yd = Q.lam + row0 + NW + NX;
  double* S_ = Q.ES(ee);
  const double* mo = Q.MO(ee);
  int fail = 0;
  // operands requested with the first batch of loads of the edge (fetch_rest(), single finite element)
  constexpr bool PF = (NI == 1 && M > 0);
  constexpr int RPL = PF ? (NW + GS_C - 1) / GS_C : 1;
  constexpr int APL = PF ? (NA + GS_C - 1) / GS_C : 1, MHL = PF ? (NX * NX + GS_C - 1) / GS_C : 1;
  double pf_xn[RPL], pf_w[RPL][DEG > 0 ? DEG : 1], pf_wend[RPL], pf_xc[RPL], pf_lam[RPL], pf_c[RPL], pf_cend[RPL];
  double pf_ltg[APL], pf_mg[RPL], pf_mh[MHL], pf_lt0 = 0.0, pf_mt0 = 0.0;
  const bool last_stage = (k == A.N - 1);
  (void)pf_xn; (void)pf_w; (void)pf_wend; (void)pf_xc; (void)pf_lam; (void)pf_c; (void)pf_cend;
  (void)pf_ltg; (void)pf_mg; (void)pf_mh; (void)pf_lt0; (void)pf_mt0; (void)last_stage;
  // the model-output record of this edge: the dense image in LDS (compact record: staged by the previous edge of this
  // wavefront / the prologue of the sweep, see stage_mo, and scattered into the image below) or global memory
  const ldsd* mol = Ld + EL_MOS;
#define MOV(i) (MO_COMPACT ? (double)mol[(i)] : mo[(i)])
#ifdef DOMPC_HOST_EMU
  if (MO_COMPACT && act) mo_expand(Ld + EL_MOS, mo, mm, lane, GS);
#endif
  (void)mm;
#ifndef DOMPC_HOST_EMU
  constexpr int PF_LINES = (MO_REC * 8 + 127) / 128, PF_N = (PF_LINES + 63) / 64;
  unsigned pf_tok[PF_N];
#pragma unroll
  for (int q = 0; q < PF_N; ++q) pf_tok[q] = 0u;
#endif
#ifndef DOMPC_HOST_EMU
  if (MO_LDS && act && staged_e != e && !(DOMPC_KO & 32)) { stage_mo(Q, e, lane, Ld); staged_e = e; }
#endif
  (void)staged_e;
  long long pc0 = prof_clock();
#if DOMPC_PROFILE
#define DOMPC_PH(i) if (T.prof && T.tid == 0) { const long long pc1 = prof_clock(); T.prof[i] += pc1 - pc0; pc0 = pc1; }
#else
#define DOMPC_PH(i)
#endif

  // ---- phase 1: zero Mx (the model-output record of eval_models is read from global memory / L2)
  if (act) {
    if (NI != 1)
      for (int i = lane; i < NW * NC; i += GS) Ld[EL_MX + i] = 0.0;
    if (NI != 1 || M == 0)      // (single element: staged below, behind the other loads of the edge)
      for (int r = lane; r < NW; r += GS) Ld[EL_T0 + r] = lam_e[r];      // multipliers of the collocation rows (dual residual)
  }
  T.gsync();
  DOMPC_PH(0)

  if (M == 0) {
    // discrete model: x_c = f(x_n,u_n); rows f - x_c with multiplier nu_e; no collocation block
    if (act) {
      const double* pt = mo + MO_PT;
      for (int a = lane; a < NX; a += GS) {
        const double r = Q.soc ? Q.c[row0 + a] : pt[a] - xc[a];
        if (!Q.soc) Q.c[row0 + a] = r;
        S_[ES_CV + a] = r;
      }
      for (int i = lane; i < NX * NA; i += GS) S_[ES_AB + i] = pt[NX + i];
      for (int i = lane; i < NA * NA; i += GS) {
        if (i / NA > i % NA) continue;             // (packed upper triangle)
        const int ip = symi(i / NA, i % NA, NA);
        double v = pt[NX + NX * NA + ip] + omh * mo[MO_LT + 1 + NA + ip];
        if (NE > 0) v += mo[MO_NL + NE + NE * NA + ip];
        S_[ES_QT + ip] = v;
      }
      for (int b = lane; b < NA; b += GS) {
        double t = 0.0;
        for (int a = 0; a < NX; ++a) t += pt[NX + a * NA + b] * nu_e[a];
        Ld[EL_RY + b] = t;          // completed in phase 7
        Ld[EL_QV + b] = 0.0;
        Ld[EL_QV + NA + b] = 0.0;
      }
    }
  } else {
    if constexpr (NI == 1) {
      // ---- phases 2-4, single finite element: [G_cc | G_y r | I] is assembled, used for the dual residual and
      // eliminated in REGISTERS, one extended column per lane - the LDS matrix of the general path does not exist
      // here (only W, w0 and G_cc^-1 are written to it afterwards for the condensing phases).
      // Single finite element: G_w = [[G_cc, 0], [E, I]] with the continuity rows E = -[D_1 I ... D_DEG I] below the
      // R x R collocation block, so only G_cc is eliminated (the continuity rows of W, w0 follow as D-weighted sums).
      // Register-resident Gauss-Jordan on the extended matrix [G_cc | G_y r | I], one COLUMN per lane (R + NA + 1 + R
      // lanes: 54 for industrial_poly): per step the pivot column is broadcast with v_readlane (it ends up in SGPRs
      // and feeds the FMAs as a scalar operand) - no LDS traffic and no barrier inside the elimination.
      // Pivoting: the natural order is tried first (the diagonal of G_cc = h J - C (x) I carries the collocation
      // coefficients C_jj) under a threshold test |a_kk| >= GJ_U max_{r >= k} |a_rk| evaluated by the lane that owns
      // column k; if any test fails, the wavefront repeats the elimination from the untouched LDS copy with partial
      // pivoting and explicit row interchanges (rare; measured: never on the BASELINE workloads).
      // (The LDS variant - column per lane re-read and re-written every step, packed pivot keys - spent two thirds of
      // its ~600 instructions per pair of steps on the redundant pivot search; this one issues ~85 per step.)
      constexpr int R = DEG * NX;
      constexpr int CPX = EF_CPX;
      // Operands of the residual rows that live outside the model-output record (iterate, multipliers; second-order
      // correction: the corrected residual), requested in one batch with the per-variable data below.  A load issued
      // between stores, or one load -> LDS store pair per loop trip, costs a full memory round trip each (stores count in
      // vmcnt on gfx9): the point-Hessian staging loop and the residual rows were 24 % of the sweep that way, the cost
      // loads behind the record stores another 10 %.
      // (indices are formed in UNSIGNED arithmetic from the lane number, byte offsets in 32 bits - ldoff(): uniform base
      //  pointer + zero-extended lane offset is an addressing mode of the global loads, a sign-extended index is not)
      const unsigned ul = (unsigned)lane, ugs = (unsigned)GS;
      const double* c_e = Q.c + row0;
      auto fetch_rest = [&]() {
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
          const unsigned it = ul + (unsigned)q * ugs, itc = it < (unsigned)NW ? it : 0u;
          const unsigned a = itc % (unsigned)NX;
          pf_xn[q] = ldoff(xn, a);
#pragma unroll
          for (int r = 1; r <= DEG; ++r) pf_w[q][r - 1] = ldoff(w, (unsigned)((r - 1) * NX) + a);
          pf_wend[q] = ldoff(w, (unsigned)((M - 1) * NX) + a);
          pf_xc[q] = ldoff(xc, a);
          pf_lam[q] = ldoff(lam_e, itc);
          pf_c[q] = Q.soc ? ldoff(c_e, itc) : 0.0;
          pf_cend[q] = Q.soc ? ldoff(c_e, (unsigned)NW + a) : 0.0;
        }
      };
      // per-variable data of the collocation unknowns (this lane's column, plus the end-point columns on the first
      // NX lanes) and the Jacobian columns: requested up front, together with the loads of the residual rows
      double vx[CPX][5], ex[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
      double nu_a = 0.0;
      if (act && (DOMPC_KO & 16)) {
#pragma unroll
        for (int q = 0; q < CPX; ++q) { vx[q][0] = 1.0; vx[q][1] = 0.0; vx[q][2] = 2.0; vx[q][3] = 1.0; vx[q][4] = 1.0; }
        ex[0] = 1.0; ex[2] = 2.0; ex[3] = 1.0; ex[4] = 1.0; nu_a = 0.5;
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
          pf_xn[q] = 1.0; pf_wend[q] = 1.0; pf_xc[q] = 1.0; pf_lam[q] = 0.5; pf_c[q] = 0.0; pf_cend[q] = 0.0;
#pragma unroll
          for (int r = 1; r <= DEG; ++r) pf_w[q][r - 1] = 1.0;
        }
#ifndef DOMPC_HOST_EMU
        if (MO_LDS && !(DOMPC_KO & 32)) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          mo_expand(Ld + EL_MOS, (const ldsd*)(Ld + EL_MOC), mm, lane, GS);
        }
#endif
      } else if (act) {
#pragma unroll
        for (int q = 0; q < CPX; ++q) {
          const unsigned cx = ul + (unsigned)q * ugs;
          const unsigned gi = cx < (unsigned)R ? cx : 0u;
          vx[q][0] = ldoff(Q.x + woff, gi); vx[q][1] = ldoff(Q.lb + woff, gi); vx[q][2] = ldoff(Q.ub + woff, gi);
          vx[q][3] = ldoff(Q.zl + woff, gi); vx[q][4] = ldoff(Q.zu + woff, gi);
        }
        if (GS > 1) {
          const unsigned gi = (unsigned)R + (ul < (unsigned)NX ? ul : 0u);
          ex[0] = ldoff(Q.x + woff, gi); ex[1] = ldoff(Q.lb + woff, gi); ex[2] = ldoff(Q.ub + woff, gi);
          ex[3] = ldoff(Q.zl + woff, gi); ex[4] = ldoff(Q.zu + woff, gi);
          nu_a = ldoff(nu_e, ul < (unsigned)NX ? ul : 0u);
        }
        fetch_rest();
#ifndef DOMPC_HOST_EMU
        if (MO_LDS && !(DOMPC_KO & 32)) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the staged record (and everything above) has landed
          mo_expand(Ld + EL_MOS, (const ldsd*)(Ld + EL_MOC), mm, lane, GS);
        }
#endif
      }
      DOMPC_PH(4)
      // residual rows (collocation, continuity, end point): computed by one lane each, written to g and staged in LDS
      // for the lane that owns the right-hand-side column; the point Hessians of the condensing phases are staged in LDS;
      // all operands were requested by fetch_rest()
      if (act) {
        if constexpr (!TILE_CONDENSE)           // (the matrix-core condensing reads the point Hessians from the record itself)
          for (int it = lane; it < NCOLL * NA * NA; it += GS)
            Ld[EL_HP + it] = MOV(MO_PT + (it / (NA * NA)) * PT_STRIDE + NX + NX * NA + symi((it % (NA * NA)) / NA, it % NA, NA));
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
          const int it = lane + q * GS;
          if (it < NW) {
            const int jj = it / NX;
            double res;
            if (jj < DEG) {
              const int j = jj + 1;
              double xp = DOMPC_C[0 * (DEG + 1) + j] * pf_xn[q];
#pragma unroll
              for (int r = 1; r <= DEG; ++r) xp += DOMPC_C[r * (DEG + 1) + j] * pf_w[q][r - 1];
              res = MOV(MO_PT + jj * PT_STRIDE + it % NX) - xp;
            } else {
              double xf = DOMPC_D[0] * pf_xn[q];
#pragma unroll
              for (int r = 1; r <= DEG; ++r) xf += DOMPC_D[r] * pf_w[q][r - 1];
              res = pf_wend[q] - xf;
            }
            if (Q.soc) res = pf_c[q];                  // (second-order correction: corrected residual instead of c(x))
            else Q.c[row0 + it] = res;
            Ld[EL_T1 + it] = res;
            Ld[EL_T0 + it] = pf_lam[q];                // multipliers of the edge's rows (dual residual)
          }
          if (it < NX) {                               // end-point rows (it = a: jj = 0, same w_end / x_c entry)
            const double ce = Q.soc ? pf_cend[q] : pf_wend[q] - pf_xc[q];
            if (!Q.soc) Q.c[row0 + NW + it] = ce;
            Ld[EL_PV + it] = ce;                       // (read back by the record stores: c~ of the edge; the pivot-row slots are free here)
          }
        }
      }
      DOMPC_PH(5)
      T.gsync();
      DOMPC_PH(6)
      if (act && !(DOMPC_KO & 1)) fail |= run_edge_factor(T, Q, e, mu, lane, GS, Ld, vx, ex, nu_a);
      T.gsync();
#ifndef DOMPC_HOST_EMU
      // the compact record of the edge this wavefront handles next: on its way into the staging buffer (free since the
      // expansion above) during the condensing phases and the stores of this edge.  Not earlier: the out-of-line
      // factorisation waits for every outstanding memory operation at its entry (calling convention).
      if (MO_LDS && e_next >= 0 && !(DOMPC_KO & 32)) { stage_mo(Q, e_next, lane, Ld); staged_e = e_next; }
#endif
    } else {
    // ---- phase 2: assemble Mx = [G_w | G_y | r_g] and the residual rows; the point Hessians needed by the
    //      condensing phases are staged in LDS with the same batch of global loads
    if (act) {
      for (int it = lane; it < NCOLL * NA * NA; it += GS)
        Ld[EL_HP + it] = mo[MO_PT + (it / (NA * NA)) * PT_STRIDE + NX + NX * NA + symi((it % (NA * NA)) / NA, it % NA, NA)];
      for (int it = lane; it < NI * (DEG + 1) * NX; it += GS) {
        const int i = it / ((DEG + 1) * NX);
        const int rr = it % ((DEG + 1) * NX);
        const int jj = rr / NX, a = rr % NX;         // jj = 0..DEG-1: collocation row j=jj+1 ; jj = DEG: continuity row
        const double* xi0 = (i == 0) ? xn : w + slot_of(i, 0) * NX;
        const int row = i * (DEG + 1) * NX + jj * NX + a;
        ldsd* Mr = Ld + EL_MX + row * NC;
        if (jj < DEG) {
          const int j = jj + 1, sl = slot_of(i, j), p = i * DEG + jj;
          const double* pt = mo + MO_PT + p * PT_STRIDE;
          double xp = DOMPC_C[0 * (DEG + 1) + j] * xi0[a];
          for (int r = 1; r <= DEG; ++r) xp += DOMPC_C[r * (DEG + 1) + j] * w[slot_of(i, r) * NX + a];
          const double res = Q.soc ? Q.c[row0 + row] : pt[a] - xp;
          if (!Q.soc) Q.c[row0 + row] = res;
          Mr[NW + NA] = res;
          for (int b = 0; b < NX; ++b) Mr[sl * NX + b] += pt[NX + a * NA + b];
          for (int b = 0; b < NU; ++b) Mr[NW + NX + b] = pt[NX + a * NA + NX + b];
          for (int r = 0; r <= DEG; ++r) {
            const double cr = DOMPC_C[r * (DEG + 1) + j];
            if (i == 0 && r == 0) Mr[NW + a] -= cr;
            else Mr[slot_of(i, r) * NX + a] -= cr;
          }
        } else {
          const int ns_ = next_slot(i);
          double xf = DOMPC_D[0] * xi0[a];
          for (int r = 1; r <= DEG; ++r) xf += DOMPC_D[r] * w[slot_of(i, r) * NX + a];
          const double res = Q.soc ? Q.c[row0 + row] : w[ns_ * NX + a] - xf;
          if (!Q.soc) Q.c[row0 + row] = res;
          Mr[NW + NA] = res;
          Mr[ns_ * NX + a] += 1.0;
          for (int r = 0; r <= DEG; ++r) {
            if (i == 0 && r == 0) Mr[NW + a] -= DOMPC_D[0];
            else Mr[slot_of(i, r) * NX + a] -= DOMPC_D[r];
          }
        }
      }
      if (!Q.soc)
        for (int a = lane; a < NX; a += GS) Q.c[row0 + NW + a] = w[(M - 1) * NX + a] - xc[a];
    }
    T.gsync();
    // ---- phase 3: dual-residual pieces that need G_w / G_y (before they are overwritten)
    if (act) {
      for (int col = lane; col < NW; col += GS) {
        double t = 0.0;
#pragma unroll 6
        for (int r = 0; r < NW; ++r) t += Ld[EL_MX + r * NC + col] * Ld[EL_T0 + r];
        if (col >= (M - 1) * NX) t += nu_e[col - (M - 1) * NX];
        const int gi = woff + col;
        const double xv = Q.x[gi], l = Q.lb[gi], u = Q.ub[gi];
        Q.gf[gi] = 0.0;
        Q.rd[gi] = t - Q.zl[gi] + Q.zu[gi];
        Ld[EL_RW + col] = t + bar_grad(xv, l, u, mu, !(Q.soc & 2));
        Ld[EL_BB + col] = bar_grad(xv, l, u, 1.0);
        Ld[EL_SG + col] = sigma_of(xv, l, u, Q.zl[gi], Q.zu[gi]);
      }
      for (int b = lane; b < NA; b += GS) {
        double t = 0.0;
#pragma unroll 6
        for (int r = 0; r < NW; ++r) t += Ld[EL_MX + r * NC + NW + b] * Ld[EL_T0 + r];
        Ld[EL_RY + b] = t;          // completed in phase 7
      }
    }
    T.gsync();
    DOMPC_PH(1)
    // in-place Gauss-Jordan inversion of [G_w | G_y | r_g] in LDS, one matrix COLUMN per lane.
    // Per step every lane loads column kk (same addresses for all lanes -> LDS broadcast) and, in the same
    // LDS round trip, its own column; the pivot row is found redundantly with a packed (|value| high word,
    // row) key - no cross-lane reduction and no row interchange (the pivot row of each column is remembered
    // and the rows are relabelled once at the end), so a step is ONE wavefront barrier and two LDS round
    // trips.  Column kk becomes the kk-th column of the inverse in place.
    // Structure: rows/columns come in groups [collocation rows of element i | continuity rows of element i]
    // (optimizer.py:943-983) and G_w is block lower-triangular in that grouping.  Pivots are searched inside
    // the group of the current column only (the diagonal blocks are the nonsingular collocation Jacobians,
    // resp. identities), which keeps the structure; the last NX columns (xkf: identity block, zero above)
    // need no elimination step at all - their inverse columns are already in place.
    {
      static_assert(NW <= 128, "row index is packed into 7 bits of the pivot key / 128-bit used mask");
      constexpr int GJ_STEPS = NW - NX;
      constexpr int EL_ROWS = (DEG + 1) * NX;
      constexpr int CPL = (NC + GS_C - 1) / GS_C;
      unsigned long long used = 0ull, used_hi = 0ull;       // (rows 64 .. 127: blocks of more than 64 unknowns, round 5)
      {
        for (int kk = 0; kk < GJ_STEPS; ++kk) {
          const int pos = kk % EL_ROWS;
          const int grp0 = kk - pos + (pos < DEG * NX ? 0 : DEG * NX);
          const int grp1 = kk - pos + (pos < DEG * NX ? DEG * NX : EL_ROWS);
          double f[NW1], bcol[CPL][NW1];
          unsigned bestkey = 0u;
          if (act) {
  #pragma unroll
            for (int r = 0; r < NW; ++r) f[r] = Ld[EL_MX + r * NC + kk];
  #pragma unroll
            for (int q = 0; q < CPL; ++q) {
              const int c = lane + q * GS;
              const int cc_ = c < NC ? c : 0;
  #pragma unroll
              for (int r = 0; r < NW; ++r) bcol[q][r] = Ld[EL_MX + r * NC + cc_];
            }
  #pragma unroll
            for (int r = 0; r < NW; ++r) {
              unsigned key = (((unsigned)(__builtin_bit_cast(unsigned long long, f[r]) >> 32)) & (NW > 64 ? 0x7fffff80u : 0x7fffffc0u)) | (unsigned)r;
              const bool taken = (r < 64) ? ((used >> (r & 63)) & 1ull) : ((used_hi >> (r & 63)) & 1ull);
              key = (r >= grp0 && r < grp1 && !taken) ? key : 0u;
              bestkey = key > bestkey ? key : bestkey;
            }
          }
          const int pv = (int)(bestkey & (NW > 64 ? 127u : 63u));
          if (pv < 64) used |= (1ull << pv); else used_hi |= (1ull << (pv - 64));
          if (act && (bestkey >> (NW > 64 ? 7 : 6)) == 0u) fail = 1;          // |pivot| < ~1e-300: singular collocation block
          if (act) {
            if (lane == 0) Ld[EL_PV + kk] = (double)pv;
            const double piv = Ld[EL_MX + pv * NC + kk];
            const double pinv = (fabs(piv) > 1e-300) ? 1.0 / piv : 1.0;
  #pragma unroll
            for (int q = 0; q < CPL; ++q) {
              const int c = lane + q * GS;
              if (c < NC) {
                const double prow = (c == kk) ? pinv : Ld[EL_MX + pv * NC + c] * pinv;
                const double keep = (c == kk) ? 0.0 : 1.0;
  #pragma unroll
                for (int r = 0; r < NW; ++r) Ld[EL_MX + r * NC + c] = fma(-f[r], prow, bcol[q][r] * keep);
                Ld[EL_MX + pv * NC + c] = prow;
              }
            }
          }
          T.gsync();
        }
      }
      // relabel: stored[p_k][c] = Ginv[k][p_c] for the inverse part, stored[p_k][c] = (Ginv B)[k][c] for the
      // right-hand sides (p_k = pivot row of column k; identity for the skipped xkf columns)
      double tmp[CPL][NW1];
      if (act) {
        for (int k2 = GJ_STEPS + lane; k2 < NW; k2 += GS) Ld[EL_PV + k2] = (double)k2;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          const int c = lane + q * GS;
          const int cc_ = c < NC ? c : 0;
#pragma unroll
          for (int r = 0; r < NW; ++r) tmp[q][r] = Ld[EL_MX + r * NC + cc_];
        }
      }
      T.gsync();
      if (act)
        for (int k2 = lane; k2 < NW; k2 += GS) Ld[EL_T0 + (int)Ld[EL_PV + k2]] = (double)k2;   // kof[row] = column it was the pivot of
      T.gsync();
      if (act) {
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          const int c = lane + q * GS;
          if (c < NC) {
            const int dst = (c < NW) ? (int)Ld[EL_PV + c] : c;
#pragma unroll
            for (int r = 0; r < NW; ++r) Ld[EL_MX + (int)Ld[EL_T0 + r] * NC + dst] = tmp[q][r];
          }
        }
      }
      T.gsync();
    }
    // now: Mx[:, :NW] = G_w^-1 ; Mx[:, NW:NW+NA] = G_w^-1 G_y = -W ; Mx[:, NW+NA] = G_w^-1 r_g = -w0
    if (act) {
      for (int it = lane; it < NW * (NA + 1); it += GS) {
        const int r = it / (NA + 1), c = it % (NA + 1);
        Ld[EL_MX + r * NC + NW + c] = -Ld[EL_MX + r * NC + NW + c];
      }
    }
    T.gsync();
    }
    DOMPC_PH(2)
    // ---- phase 5 (device, single finite element, <= 16 stage variables): condensing on the matrix cores with
    //      register-resident tiles.  Per collocation point p the stage variables are (x_p; u) = Z_p y + z0_p with
    //      Z_p = [W_p; E_u], z0_p = (w0_p; 0), so
    //          Q~ = omega H_l + H_nl + sum_p Z_p'(H_p + Sigma_p) Z_p + W_k' Sigma_k W_k          (k: end-point slot)
    //          q~ = sum_p Z_p'((H_p + Sigma_p) z0_p + rw_p) + W_k'(Sigma_k w0_k + rw_k)
    //      - 38 MFMAs instead of the LDS-staged products of the generic path below (H_ww W, H_uw W, W'T1, ...).
    if constexpr (TILE_CONDENSE) {
#ifndef DOMPC_HOST_EMU
      if (act && !(DOMPC_KO & 2)) {
        DOMPC_PRIO_UP();
        constexpr int KB_A = (NA + 3) / 4, KB_X = (NX + 3) / 4;
        const int g = lane >> 4, j = lane & 15;
        auto Wm = [&](int row, int col) -> double { return Ld[EL_MX + row * MX_LD + MX_W + col]; };
        // (NA + 2 <= 16: the vector parts ride in the spare columns of the matrix tiles - column NA: (H + Sigma) z0 + r_w
        //  -> q~, column NA + 1: b -> W'b - so a point costs 8 MFMAs instead of 16, the end-point slot 3 instead of 6)
        constexpr bool VCOL = NA + 2 <= 16;
        d4 QTt, qv0 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {                  // stage-cost and nl_cons Hessians (packed in the model-output record)
          const int i = g + 4 * r;
          const bool in = i < NA && j < NA;
          const int ip = in ? symi(i, j, NA) : 0;
          double v = omh * MOV(MO_LT + 1 + NA + ip);
          if (NE > 0) v += MOV(MO_NL + NE + NE * NA + ip);
          QTt[r] = in ? v : 0.0;
        }
#pragma unroll
        for (int p = 0; p < NCOLL; ++p) {              // (NI == 1: point p lives in slot p)
          d4 Z, z0, H, rwv;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = g + 4 * r;
            const int row = p * NX + (i < NX ? i : 0);
            const double wv = Wm(row, j <= NA ? j : 0), w0v = VCOL ? 0.0 : Wm(row, NA);
            const double hv = MOV(MO_PT + p * PT_STRIDE + NX + NX * NA + symi(i < NA ? i : 0, j < NA ? j : 0, NA));
            const double sg = Ld[EL_SG + row] + Q.dsw, rw = Ld[EL_RW + row], bb = Ld[EL_BB + row];
            Z[r] = (i < NX) ? (j < NA + (VCOL ? 1 : 0) ? wv : 0.0) : ((i < NA && j == i) ? 1.0 : 0.0);      // VCOL: [W_p | w0_p]
            z0[r] = (j == 0 && i < NX) ? w0v : 0.0;
            H[r] = (i < NA && j < NA) ? hv + ((i == j && i < NX) ? sg : 0.0) : 0.0;
            const int jv = VCOL ? NA : 0;
            rwv[r] = (i < NX) ? (j == jv ? rw : (j == jv + 1 ? bb : 0.0)) : 0.0;      // (second vector column: the part of the gradient that is linear in mu -> W'b)
          }
          if constexpr (VCOL) {
            const d4 HZ = tile_mul<KB_A>(H, Z) + rwv;
            QTt += tile_mul<KB_A>(Z, HZ);
          } else {
            const d4 HZ = tile_mul<KB_A>(H, Z);
            const d4 hz0 = tile_mul<KB_A>(H, z0) + rwv;
            QTt += tile_mul<KB_A>(Z, HZ);
            qv0 += tile_mul<KB_A>(Z, hz0);
          }
        }
        {
          d4 Wk, SWk, sv0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = g + 4 * r;
            const int row = (M - 1) * NX + (i < NX ? i : 0);
            const double wv = Wm(row, j <= NA ? j : 0), w0v = Wm(row, NA);
            const double sg = Ld[EL_SG + row] + Q.dsw, rw = Ld[EL_RW + row], bb = Ld[EL_BB + row];
            Wk[r] = (i < NX && j < NA + (VCOL ? 1 : 0)) ? wv : 0.0;
            const double vec0 = sg * w0v + rw;
            SWk[r] = (i < NX) ? (j < NA ? sg * wv : ((VCOL && j == NA) ? vec0 : ((VCOL && j == NA + 1) ? bb : 0.0))) : 0.0;
            sv0[r] = (i < NX) ? (j == 0 ? vec0 : (j == 1 ? bb : 0.0)) : 0.0;
          }
          QTt += tile_mul<KB_X>(Wk, SWk);
          if constexpr (!VCOL) qv0 += tile_mul<KB_X>(Wk, sv0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = g + 4 * r;
          if (i <= j && j < NA) S_[ES_QT + symi(i, j, NA)] = QTt[r];
          // q~ and W'b: stored by phase 7
          if constexpr (VCOL) { if (i < NA && (j == NA || j == NA + 1)) Ld[EL_QV + (j - NA) * NA + i] = QTt[r]; }
          else { if (i < NA && j < 2) Ld[EL_QV + j * NA + i] = qv0[r]; }
        }
        DOMPC_PRIO_DOWN();
      }
#endif
      DOMPC_PH(7)
    } else {
    // ---- phase 5 (generic): T1 = Hww W, t0 = Hww w0, U1 = Huw W, u0 = Huw w0   (Hww = blockdiag(Hxx_p) + Sigma_w)
    //      (stage-cost / nl_cons Hessian entries for phase 6 are requested now, consumed there)
    constexpr int QPL = (NA * NA + GS_C - 1) / GS_C;
    double qlt[QPL], qnl[QPL];
#pragma unroll
    for (int q = 0; q < QPL; ++q) {
      const int it = lane + q * GS;
      const int itc = it < NA * NA ? it : 0;
      const int ip = symi(itc / NA, itc % NA, NA);
      qlt[q] = act ? MOV(MO_LT + 1 + NA + ip) : 0.0;
      qnl[q] = (act && NE > 0) ? MOV(MO_NL + NE + NE * NA + ip) : 0.0;
    }
    if (act) {
      for (int it = lane; it < NW * (NA + 1); it += GS) {
        const int row = it / (NA + 1), b = it % (NA + 1);
        double t = (Ld[EL_SG + row] + Q.dsw) * Ld[EL_MX + row * MX_LD + MX_W + b];
        if (b == NA) {            // the w0 column: small mat-vec on the vector units
          const int sl = row / NX, a = row % NX;
          const int p = point_of_slot(sl);
          if (p >= 0) {
            const ldsd* Hp = Ld + EL_HP + p * NA * NA;
#pragma unroll
            for (int a2 = 0; a2 < NX; ++a2) t += Hp[a * NA + a2] * Ld[EL_MX + (sl * NX + a2) * MX_LD + MX_W + NA];
          }
          Ld[EL_T0 + row] = t;
        } else {
          Ld[EL_T1 + row * NA + b] = t;
        }
      }
      for (int it = lane; it < NU * (NA + 1); it += GS) {
        const int ub = it / (NA + 1), b = it % (NA + 1);
        double t = 0.0;
        for (int p = 0; p < NCOLL; ++p) {
          const int sl = slot_of(p / DEG, p % DEG + 1);
          const ldsd* Hp = Ld + EL_HP + p * NA * NA;
#pragma unroll
          for (int a = 0; a < NX; ++a) t += Hp[a * NA + NX + ub] * Ld[EL_MX + (sl * NX + a) * MX_LD + MX_W + b];
        }
        Ld[EL_U1 + (b < NA ? ub * NA + b : NU * NA + ub)] = t;
      }
      for (int it = lane; it < NU * NU; it += GS) {
        double h = 0.0;
        for (int p = 0; p < NCOLL; ++p) h += Ld[EL_HP + p * NA * NA + (NX + it / NU) * NA + NX + it % NU];
        Ld[EL_HUU + it] = h;
      }
    }
    T.gsync();
    if (act) {
      // T1[slot rows] += Hxx_p * W[slot rows]   (matrix cores)
      for (int p = 0; p < NCOLL; ++p) {
        const int sl = slot_of(p / DEG, p % DEG + 1);
        gmm(lane, GS, NX, NA, NX, (double*)(Ld + EL_HP + p * NA * NA), NA, 1,
            (double*)(Ld + EL_MX + (sl * NX) * MX_LD + MX_W), MX_LD, 1, 1.0, (double*)(Ld + EL_T1 + sl * NX * NA), NA);
      }
    }
    T.gsync();
    if (act) {
      // W'T1 and W'W  (13x30 * 30x13 on the matrix cores)
      gmm(lane, GS, NA, NA, NW, (double*)(Ld + EL_MX + MX_W), 1, MX_LD, (double*)(Ld + EL_T1), NA, 1, 0.0, (double*)(Ld + EL_QT), NA);
    }
    T.gsync();
    if (act) {
#pragma unroll
      for (int qi = 0; qi < QPL; ++qi) {
        const int it = lane + qi * GS;
        if (it < NA * NA) {
          const int a1 = it / NA, b = it % NA;
          double q = omh * qlt[qi] + Ld[EL_QT + it];
          if (NE > 0) q += qnl[qi];
          if (a1 >= NX && b >= NX) q += Ld[EL_HUU + (a1 - NX) * NU + (b - NX)];
          if (a1 >= NX) q += Ld[EL_U1 + (a1 - NX) * NA + b];
          if (b >= NX) q += Ld[EL_U1 + (b - NX) * NA + a1];
          if (a1 <= b) S_[ES_QT + symi(a1, b, NA)] = q;
        }
      }
      for (int a1 = lane; a1 < NA; a1 += GS) {
        double q = 0.0;
        for (int row = 0; row < NW; ++row) q += Ld[EL_MX + row * MX_LD + MX_W + a1] * (Ld[EL_RW + row] + Ld[EL_T0 + row]);
        if (a1 >= NX) q += Ld[EL_U1 + NU * NA + a1 - NX];
        Ld[EL_QV + a1] = q;                   // stored by phase 7 together with r_y
        double qb = 0.0;                      // W'b: the part of q~ that is linear in mu (refresh_mu)
        for (int row = 0; row < NW; ++row) qb += Ld[EL_MX + row * MX_LD + MX_W + a1] * Ld[EL_BB + row];
        Ld[EL_QV + NA + a1] = qb;
      }
    }
    }
    // ---- phase 6: condensed blocks to the shared per-edge record; data for the forward pass
    if constexpr (PF) {
      // the cost pieces of the record that phases 6-7 still need, taken out before the staging area is handed to the next edge
      if (act) {
#pragma unroll
        for (int q = 0; q < APL; ++q) {
          const int a = lane + q * GS;
          pf_ltg[q] = MOV(MO_LT + 1 + (a < NA ? a : 0));
        }
        pf_lt0 = MOV(MO_LT);
        if (last_stage) {
#pragma unroll
          for (int q = 0; q < RPL; ++q) {
            const int a = lane + q * GS;
            pf_mg[q] = MOV(MO_MT + 1 + (a < NX ? a : 0));
          }
#pragma unroll
          for (int q = 0; q < MHL; ++q) {
            const int a = lane + q * GS, ac = a < NX * NX ? a : 0;
            pf_mh[q] = MOV(MO_MT + 1 + NX + symi(ac / NX, ac % NX, NX));
          }
          pf_mt0 = MOV(MO_MT);
        }
      }
    }
#ifndef DOMPC_HOST_EMU
    // (device variants without the compact record) touch the model-output record of the edge this wavefront handles next (one dword per
    // 128-byte line): by the time its assembly starts the lines sit in L2 instead of HBM.  The values are consumed (never
    // true) at the end of the function so that the loads stay where they are.
    if constexpr (!MO_LDS) {
#pragma unroll
      for (int q = 0; q < PF_N; ++q) {
        const int line = lane + 64 * q;
        pf_tok[q] = (e_next >= 0 && line < PF_LINES)
                        ? *((const volatile unsigned*)((const char*)Q.MO(e_next) + (int64_t)line * 128)) : 0u;
      }
    }
#endif
    if (act && !(DOMPC_KO & 4)) {
      for (int it = lane; it < NX * (NA + 1); it += GS) {
        const int a = it / (NA + 1), b = it % (NA + 1);
        const double v = Ld[EL_MX + ((M - 1) * NX + a) * MX_LD + MX_W + b];
        if (b < NA) S_[ES_AB + a * NA + b] = v;
        else if (PF) S_[ES_CV + a] = v + Ld[EL_PV + a];
        else S_[ES_CV + a] = v + (Q.soc ? Q.c[row0 + NW + a] : w[(M - 1) * NX + a] - xc[a]);
      }
      // forward-pass data (interleaved per-edge workspace)
      if (NI != 1)       // (single element: G_cc^-1 went to the record straight from the registers)
        for (int it = lane; it < LU_N * LU_N; it += GS) Q.EW(e, EW_LU + it) = Ld[EL_MX + (it / LU_N) * NC + it % LU_N];
      for (int r = lane; r < NW; r += GS) {
        Q.EW(e, EW_SIGW + r) = Ld[EL_SG + r] + Q.dsw;
        Q.EW(e, EW_RW + r) = Ld[EL_RW + r];
      }
    }
  }
  T.gsync();
  // ---- phase 7: stage cost / terminal cost / nl_cons shares (few values: lanes 0..)
  if (RT_CUSTOM) {                         // user-defined rterm: one lane evaluates it (value, gradient, Hessian) into LDS
    if (act && lane == 0) edge_rterm_eval(Q, e, Ld + EL_RT);
    T.gsync();
    if (act) edge_rterm_store(Ld + EL_RT, S_, lane, GS);
  }
  if (act && !(DOMPC_KO & 4)) {
    if constexpr (PF) {                    // (operands in registers since the first load batch of the edge)
#pragma unroll
      for (int q = 0; q < APL; ++q) {
        const int a = lane + q * GS;
        if (a < NA) {
          const double grt = RT_CUSTOM ? (double)Ld[EL_RT + 1 + a] : 0.0;       // d rterm / d (x_n, u_n)
          double r = Ld[EL_RY + a] + om * pf_ltg[q] + grt;
          if (NE > 0)
            for (int i = 0; i < NE; ++i) r += MOV(MO_NL + NE + i * NA + a) * yd[i] * Q.sgn[e * NE1 + i];
          S_[ES_GFY + a] = om * pf_ltg[q] + grt;
          S_[ES_RY + a] = r;
          S_[ES_QV + a] = Ld[EL_QV + a] + r;
          S_[ES_QVB + a] = Ld[EL_QV + NA + a];
        }
      }
      if (last_stage) {
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
          const int a = lane + q * GS;
          if (a < NX) S_[ES_MG + a] = om * pf_mg[q];
        }
#pragma unroll
        for (int q = 0; q < MHL; ++q) {
          const int a = lane + q * GS;
          if (a < NX * NX) S_[ES_MH + a] = omh * pf_mh[q];
        }
      }
    } else {
    for (int a = lane; a < NA; a += GS) {
      const double grt = RT_CUSTOM ? (double)Ld[EL_RT + 1 + a] : 0.0;           // d rterm / d (x_n, u_n)
      double r = Ld[EL_RY + a] + om * mo[MO_LT + 1 + a] + grt;
      if (NE > 0)
        for (int i = 0; i < NE; ++i) r += mo[MO_NL + NE + i * NA + a] * yd[i] * Q.sgn[e * NE1 + i];
      S_[ES_GFY + a] = om * mo[MO_LT + 1 + a] + grt;
      S_[ES_RY + a] = r;
      S_[ES_QV + a] = Ld[EL_QV + a] + r;
      S_[ES_QVB + a] = Ld[EL_QV + NA + a];
    }
    if (k == A.N - 1) {
      for (int a = lane; a < NX; a += GS) S_[ES_MG + a] = om * mo[MO_MT + 1 + a];
      for (int a = lane; a < NX * NX; a += GS) S_[ES_MH + a] = omh * mo[MO_MT + 1 + NX + symi(a / NX, a % NX, NX)];
    }
    }
    if (lane == 0) {
      double obj = PF ? om * pf_lt0 : om * mo[MO_LT];
      if (k == A.N - 1) obj += PF ? om * pf_mt0 : om * mo[MO_MT];
      if (RT_CUSTOM) obj += Ld[EL_RT];
      if (NE > 0) {
        const double* eps = (NSE > 0) ? Q.x + eps_off_n : nullptr;
        for (int i = 0; i < NE; ++i) {
          double d = MOV(MO_NL + i);
          if (nl_slack(i) >= 0) d -= eps[nl_slack(i)];
          const int si = e * NE1 + i;
          d *= Q.sgn[si];
          const double sv = Q.s[si], l = Q.sl[si], u = Q.su[si];
          const double rdn = Q.soc ? Q.c[row0 + NW + NX + i] : d - sv;
          if (!Q.soc) Q.c[row0 + NW + NX + i] = rdn;
          S_[ES_RDN + i] = rdn;
          S_[ES_SIGS + i] = sigma_of(sv, l, u, Q.zsl[si], Q.zsu[si]);
          S_[ES_RSN + i] = -yd[i] + bar_grad(sv, l, u, mu);
        }
        for (int q = 0; q < NSE; ++q) obj += Q.sf * DOMPC_EPS_PEN[q] * eps[q];
      }
      S_[ES_OBJ] = obj;
    }
    if (NE > 0)
      for (int it = lane; it < NE * NA; it += GS) Q.EW(e, EW_JD + it) = MOV(MO_NL + NE + it) * Q.sgn[e * NE1 + it / NA];
  }
  T.gsync();
#ifndef DOMPC_HOST_EMU
  {
    unsigned acc = 0u;
#pragma unroll
    for (int q = 0; q < PF_N; ++q) acc |= pf_tok[q] == 0x7ff8deadu ? 1u : 0u;
    if (acc && mu < 0.0) fail = 1;
  }
#endif
  DOMPC_PH(3)
#undef DOMPC_PH
#undef MOV
  return fail;
}

// ================================================================================================
// Gradient / dual-residual assembly for the variables owned by node n (x_n, u_n, eps_n), in two parts so that
// the child-dependent sums of a cut parent (tree sharding) can be exchanged between the ranks:
//   assemble_children: sums over the child edges e with take(e):  [gx | rx | gu | ru | child rterm | r_eps]
//   assemble_finish:   adds the node's own terms and writes gf / rd.
constexpr int ASM_N = 2 * NX + 3 * NU + NS;
// exchange buffer layout of a sharded problem (doubles):
//   [reduction table W x RED_MAX | cut parents' assembly sums + W sweep flags | cut Riccati pass 1 | pass 2 + W flags]
constexpr int CUT1 = 2 * (NYT * NYT + NYT);       // QO, QOV, QF, QFV
constexpr int CUT2 = NA * NA + NA;                // closed-loop value-function share PN, PNV
DOMPC_DEV inline int x_asm(const KArgs& A) { return A.shard_world * RED_MAX; }
DOMPC_DEV inline int x_c1(const KArgs& A) { return x_asm(A) + A.n_cut * ASM_N + A.shard_world; }
DOMPC_DEV inline int x_c2(const KArgs& A) { return x_c1(A) + A.n_cut * CUT1; }
DOMPC_DEV inline void assemble_children(const Prob& Q, int n, bool counted_only, double* out) {
  const KArgs& A = *Q.A;
  const int cs = A.node_child_start[n], cc = A.node_child_count[n];
  const int uo = A.node_u_off[n];
  for (int i = 0; i < ASM_N; ++i) out[i] = 0.0;
  for (int j = 0; j < cc; ++j) {
    const int e = cs + j;
    if (counted_only && !sh_cnt(A, mk_e(A, e))) continue;
    const double* S_ = Q.ES(e);
    for (int a = 0; a < NX; ++a) { out[a] += S_[ES_GFY + a]; out[NX + a] += S_[ES_RY + a]; }
    for (int i = 0; i < NU; ++i) { out[2 * NX + i] += S_[ES_GFY + NX + i]; out[2 * NX + NU + i] += S_[ES_RY + NX + i]; }
    const int cn = A.edge_child[e];
    if (A.node_u_off[cn] >= 0) {                   // the child's rterm w.r.t. its u_prev = u_n
      if (RT_CUSTOM) {                             // (user-defined: one term per edge leaving the child)
        for (int j2 = 0; j2 < A.node_child_count[cn]; ++j2) {
          const double* S2 = Q.ES(A.node_child_start[cn] + j2);
          for (int i = 0; i < NU; ++i) out[2 * NX + 2 * NU + i] += S2[ES_RTUP + i];
        }
      } else {
        const double rwc = node_rweight(Q, cn);
        for (int i = 0; i < NU; ++i)
          out[2 * NX + 2 * NU + i] -= 2.0 * rwc * DOMPC_RTERM[i] * (Q.x[A.node_u_off[cn] + i] - Q.x[uo + i]);
      }
    }
    if (NS > 0) {
      const double* yd = Q.lam + A.edge_row0[e] + NW + NX;
      for (int q = 0; q < NS; ++q)
        for (int i = 0; i < NE; ++i)
          if (nl_slack(i) == q) out[2 * NX + 3 * NU + q] -= yd[i] * Q.sgn[e * NE1 + i];
    }
  }
}
DOMPC_DEV inline void assemble_finish(const Prob& Q, int n, const double* in) {
  const KArgs& A = *Q.A;
  const int cc = A.node_child_count[n];
  const int xo = A.node_x_off[n];
  const int ie = A.node_in_edge[n];
  for (int a = 0; a < NX; ++a) {
    double gx = in[a], rx = in[NX + a];
    if (ie >= 0) {
      rx -= Q.lam[A.edge_row0[ie] + NW + a];
      if (cc == 0) { const double mg = Q.ES(ie)[ES_MG + a]; gx += mg; rx += mg; }
    } else if (FREE_ROOT) {
      const double ga = Q.ND(0)[ND_AT + 1 + a];      // free initial state: gradient of the arrival cost
      gx += ga; rx += ga;
    } else {
      rx += Q.lam[a];
    }
    Q.gf[xo + a] = gx;
    Q.rd[xo + a] = rx - Q.zl[xo + a] + Q.zu[xo + a];
  }
  if (cc == 0) return;
  const int uo = A.node_u_off[n];
  double tmp[NU];
  const double* up = uprev_ptr(Q, n, Q.x, tmp);
  const double rw = node_rweight(Q, n);
  for (int i = 0; i < NU; ++i) {
    const double rt = (RT_CUSTOM ? 0.0 : 2.0 * rw * DOMPC_RTERM[i] * (Q.x[uo + i] - up[i])) + in[2 * NX + 2 * NU + i];    // (user-defined rterm: own share is in the edges' GFY / RY)
    Q.gf[uo + i] = in[2 * NX + i] + rt;
    Q.rd[uo + i] = in[2 * NX + NU + i] + rt - Q.zl[uo + i] + Q.zu[uo + i];
  }
  if (NS > 0) {
    const int eo = A.node_eps_off[n];
    for (int q = 0; q < NS; ++q) {
      const double g = cc * Q.sf * DOMPC_EPS_PEN[q];
      Q.gf[eo + q] = g;
      Q.rd[eo + q] = g + in[2 * NX + 3 * NU + q] - Q.zl[eo + q] + Q.zu[eo + q];
    }
  }
}
DOMPC_PHASE void assemble_node(const Prob& Q, int n) {
  double t[ASM_N];
  assemble_children(Q, n, false, t);
  assemble_finish(Q, n, t);
}
// One problem spread over several workgroups (wide mode: B <= 64) has thousands of threads for a few hundred nodes / edges: the thread-per-node
// and thread-per-edge loops of the sweep and of the line search then run as thread-per-ENTRY loops (same arithmetic per entry, same order of
// the sums: bitwise the same results).  -DDOMPC_FINE_ITEMS=1: everywhere (test of these paths on the host emulation).
#ifndef DOMPC_FINE_ITEMS
#define DOMPC_FINE_ITEMS 0
#endif
DOMPC_DEV inline bool fine_items(const Thr& T, const KArgs& A) { return DOMPC_FINE_ITEMS >= 0 && ((WIDE_OK && T.nwg > 1) || DOMPC_FINE_ITEMS > 0) && !sh_on(A); }      // (-1: compiled out, A/B measurements)
// assemble_node for ONE variable of node n: j < NX state, < NX + NU input, else slack entry
DOMPC_DEV inline void assemble_entry(const Prob& Q, int n, int j) {
  const KArgs& A = *Q.A;
  const int cs = A.node_child_start[n], cc = A.node_child_count[n];
  if (j < NX) {
    const int a = j, xo = A.node_x_off[n], ie = A.node_in_edge[n];
    double gx = 0.0, rx = 0.0;
    for (int c = 0; c < cc; ++c) { const double* S_ = Q.ES(cs + c); gx += S_[ES_GFY + a]; rx += S_[ES_RY + a]; }
    if (ie >= 0) {
      rx -= Q.lam[A.edge_row0[ie] + NW + a];
      if (cc == 0) { const double mg = Q.ES(ie)[ES_MG + a]; gx += mg; rx += mg; }
    } else if (FREE_ROOT) {
      const double ga = Q.ND(0)[ND_AT + 1 + a];
      gx += ga; rx += ga;
    } else {
      rx += Q.lam[a];
    }
    Q.gf[xo + a] = gx;
    Q.rd[xo + a] = rx - Q.zl[xo + a] + Q.zu[xo + a];
    return;
  }
  if (cc == 0) return;
  if (j < NX + NU) {
    const int i = j - NX, uo = A.node_u_off[n];
    double gu = 0.0, ru = 0.0, crt = 0.0;
    for (int c = 0; c < cc; ++c) {
      const int e = cs + c;
      const double* S_ = Q.ES(e);
      gu += S_[ES_GFY + NX + i]; ru += S_[ES_RY + NX + i];
      const int cn = A.edge_child[e];
      if (A.node_u_off[cn] >= 0) {
        if (RT_CUSTOM) {
          for (int j2 = 0; j2 < A.node_child_count[cn]; ++j2) crt += Q.ES(A.node_child_start[cn] + j2)[ES_RTUP + i];
        } else {
          crt -= 2.0 * node_rweight(Q, cn) * DOMPC_RTERM[i] * (Q.x[A.node_u_off[cn] + i] - Q.x[uo + i]);
        }
      }
    }
    double tmp[NU];
    const double* up = uprev_ptr(Q, n, Q.x, tmp);
    const double rt = (RT_CUSTOM ? 0.0 : 2.0 * node_rweight(Q, n) * DOMPC_RTERM[i] * (Q.x[uo + i] - up[i])) + crt;
    Q.gf[uo + i] = gu + rt;
    Q.rd[uo + i] = ru + rt - Q.zl[uo + i] + Q.zu[uo + i];
    return;
  }
  if (NS > 0) {
    const int q = j - NX - NU, eo = A.node_eps_off[n];
    double r = 0.0;
    for (int c = 0; c < cc; ++c) {
      const int e = cs + c;
      const double* yd = Q.lam + A.edge_row0[e] + NW + NX;
      for (int i = 0; i < NE; ++i)
        if (nl_slack(i) == q) r -= yd[i] * Q.sgn[e * NE1 + i];
    }
    const double g = cc * Q.sf * DOMPC_EPS_PEN[q];
    Q.gf[eo + q] = g;
    Q.rd[eo + q] = g + r - Q.zl[eo + q] + Q.zu[eo + q];
  }
}

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

// Forward sweep: steps for node variables, then per edge the collocation steps and multipliers.
// One group of lanes per node (level by level), then one group per edge.
#ifndef DOMPC_ADJ_REFINE
#define DOMPC_ADJ_REFINE 1            // adjoint recovery of the continuity multipliers (0: the steps of round 4, d nu = P dx + p everywhere)
#endif
#ifndef DOMPC_ADJ_MU
#define DOMPC_ADJ_MU 10.0             // used from mu <= DOMPC_ADJ_MU * tol on: the last one or two levels of the barrier parameter (default tolerance:
#endif                                // 2.5e-9 and 9.1e-10), where Sigma reaches 1e9 ... 1e11 (measured: the same iteration counts from 1e-8 to 1e-3, half the cost of 1e-5)
// (its own instantiation of the forward pass - on the device its own outlined phase: the per-edge part of the other one keeps its registers)
DOMPC_DEV inline bool forward_adjoint(const Prob& Q, double mu) {
  constexpr bool ok = DOMPC_ADJ_REFINE && NI == 1 && M > 0 && DEG > 0 && !DENSE_EDGE && !RT_CUSTOM && !FREE_ROOT && !EPS_GLOBAL;
  return ok && !sh_on(*Q.A) && !(Q.soc & 2) && mu > 0.0 && mu <= DOMPC_ADJ_MU * Q.A->opt.tol;
}
template <bool ADJ>
DOMPC_PHASE void riccati_forward_t(const Thr& T, const Prob& Q, double mu, double delta) {
  const KArgs& A = *Q.A;
  const int GS = T.gs, ng = T.nt / GS, gid = group_index(T.tid, GS), lane = T.tid % GS;
  ldsd* Ld = T.edge_lds + (int64_t)(T.ltid / GS) * EL_SIZE;
  // operands of a chain-node step, staged in LDS: own gains [K | kv], the child edge's [A B | c], the first NX rows of
  // the child's value function [P_c | p_c]
  constexpr int FW_K = NV * NA + NV, FW_AB = NX * NA + NX, FW_N = FW_K + 2 * FW_AB;
  constexpr int FW_PL = (FW_N + GS_C - 1) / GS_C;
  constexpr int RF_DX = 0, RF_DV = RF_DX + NA, RF_DY = RF_DV + NV, RF_DNU = RF_DY + NA, RF_DW = RF_DNU + NX,
                RF_RHS = RF_DW + NW1, RF_G = RF_RHS + NW1, RF_DXN = RF_G + NW1, RF_IN = RF_DXN + NA;
  static_assert(RF_IN + FW_N <= EL_SIZE, "forward working set must fit the per-group LDS region");
  static_assert(RF_IN + FW_N <= RF_EW, "the staged edge records start behind the step vectors and chain-step operands");
  long long pc0 = prof_clock();
#if DOMPC_PROFILE
#define DOMPC_PF(i) if (T.prof && T.tid == 0) { const long long pc1 = prof_clock(); T.prof[i] += pc1 - pc0; pc0 = pc1; }
#else
#define DOMPC_PF(i)
#endif
  (void)pc0;
  // root
  if (T.tid == 0) {
    double* Nd = Q.ND(0);
    const int xo = A.node_x_off[0];
    if (FREE_ROOT) {                                         // (the step of the free initial state was formed at the end of the backward pass)
      for (int a = 0; a < NX; ++a) Q.dx[xo + a] = Nd[ND_DXT + a];
    } else {
      for (int a = 0; a < NX; ++a) { Nd[ND_DXT + a] = -Q.c[a]; Q.dx[xo + a] = -Q.c[a]; }
    }
    for (int a = NX; a < NA; ++a) Nd[ND_DXT + a] = 0.0;
  }
  T.sync();
  // node steps: dv = K dx~ + kv, children dx~ = Atilde [dx~; dv] + c~.  Branching stages level by level with
  // a barrier; below the robust horizon each group walks its scenario chain downwards with dx~ kept in LDS.
  auto node_step = [&](int n) {                         // generic (any number of children; operands from global memory)
    const double* Nd = Q.ND(n);
    for (int a = lane; a < NA; a += GS) Ld[RF_DX + a] = Nd[ND_DXT + a];
    T.gsync();
    for (int i = lane; i < NV; i += GS) {
      double t = Nd[ND_KV + i];
#pragma unroll
      for (int a = 0; a < NA; ++a) t += Nd[ND_K + i * NA + a] * Ld[RF_DX + a];
      Ld[RF_DV + i] = t;
      if (i < NU) Q.dx[A.node_u_off[n] + i] = t;
      else Q.dx[A.node_eps_off[n] + i - NU] = t;
    }
    T.gsync();
    const int cs = A.node_child_start[n], cc = A.node_child_count[n];
    for (int it = lane; it < cc * NA; it += GS) {
      const int e = cs + it / NA, a = it % NA, cn = A.edge_child[e];
      if (!mk_e(A, e)) continue;                       // another rank's sub-tree
      const double* S_ = Q.ES(e);
      double t;
      if (a < NX) {
        t = S_[ES_CV + a];
#pragma unroll
        for (int b = 0; b < NX; ++b) t += S_[ES_AB + a * NA + b] * Ld[RF_DX + b];
#pragma unroll
        for (int b = 0; b < NU; ++b) t += S_[ES_AB + a * NA + NX + b] * Ld[RF_DV + b];
        Q.dx[A.node_x_off[cn] + a] = t;
      } else {
        t = Ld[RF_DV + a - NX];
      }
      Q.ND(cn)[ND_DXT + a] = t;
    }
    T.gsync();
  };
  // chain node (one child): operands requested one node ahead (load_step), staged through LDS; also forms the
  // multiplier step of the child's incoming continuity rows  d nu = P_c dx~_c + p_c  (x rows)
  auto load_step = [&](int n, double (&v)[FW_PL]) {
    const int e = A.node_child_start[n];
    const double *Nd = Q.ND(n), *S_ = Q.ES(e), *Nc = Q.ND(A.edge_child[e]);
#pragma unroll
    for (int q = 0; q < FW_PL; ++q) {
      const int i = lane + q * GS;
      double x = 0.0;
      if (i < NV * NA) x = Nd[ND_K + i];
      else if (i < FW_K) x = Nd[ND_KV + i - NV * NA];
      else if (i < FW_K + NX * NA) x = S_[ES_AB + i - FW_K];
      else if (i < FW_K + FW_AB) x = S_[ES_CV + i - FW_K - NX * NA];
      else if (i < FW_K + FW_AB + NX * NA) x = Nc[ND_P + i - FW_K - FW_AB];
      else if (i < FW_N) x = Nc[ND_PV + i - FW_K - FW_AB - NX * NA];
      v[q] = x;
    }
  };
  auto chain_step = [&](int n, const double (&v)[FW_PL]) {      // dx~ of node n is in Ld[RF_DX]
    const int e = A.node_child_start[n], cn = A.edge_child[e];
#pragma unroll
    for (int q = 0; q < FW_PL; ++q) {
      const int i = lane + q * GS;
      if (i < FW_N) Ld[RF_IN + i] = v[q];
    }
    T.gsync();
    const ldsd *K_ = Ld + RF_IN, *KV_ = K_ + NV * NA, *AB_ = Ld + RF_IN + FW_K, *CV_ = AB_ + NX * NA,
               *PC_ = Ld + RF_IN + FW_K + FW_AB, *PV_ = PC_ + NX * NA;
    for (int i = lane; i < NV; i += GS) {
      double t = KV_[i];
#pragma unroll
      for (int a = 0; a < NA; ++a) t += K_[i * NA + a] * Ld[RF_DX + a];
      Ld[RF_DV + i] = t;
      if (i < NU) Q.dx[A.node_u_off[n] + i] = t;
      else Q.dx[A.node_eps_off[n] + i - NU] = t;
    }
    T.gsync();
    for (int a = lane; a < NA; a += GS) {
      double t;
      if (a < NX) {
        t = CV_[a];
#pragma unroll
        for (int b = 0; b < NX; ++b) t += AB_[a * NA + b] * Ld[RF_DX + b];
#pragma unroll
        for (int b = 0; b < NU; ++b) t += AB_[a * NA + NX + b] * Ld[RF_DV + b];
        Q.dx[A.node_x_off[cn] + a] = t;
      } else {
        t = Ld[RF_DV + a - NX];
      }
      Q.ND(cn)[ND_DXT + a] = t;
      Ld[RF_DXN + a] = t;
    }
    T.gsync();
    for (int a = lane; a < NX; a += GS) {
      double t = PV_[a];
#pragma unroll
      for (int b = 0; b < NA; ++b) t += PC_[a * NA + b] * Ld[RF_DXN + b];
      Q.dlam[A.edge_row0[e] + NW + a] = t;
    }
    for (int a = lane; a < NA; a += GS) Ld[RF_DX + a] = Ld[RF_DXN + a];
    T.gsync();
  };
  const int cl = A.chain_level < A.N ? A.chain_level : A.N;
  for (int k = 0; k < cl; ++k) {
    const int n0 = A.level_node_start[k], n1 = A.level_node_start[k + 1];
    for (int n = n0 + gid; n < n1; n += ng)
      if (mk_n(A, n)) node_step(n);
    T.sync();
  }
#ifndef DOMPC_HOST_EMU
#ifndef DOMPC_FW4
#define DOMPC_FW4 1                 // chain walk of the forward pass: four scenario chains per wavefront (0: one)
#endif
  // Chain walk, FOUR scenario chains per wavefront: a chain step keeps at most NA (<= 16) lanes busy and is a sequence of four LDS round
  // trips with dependent sums in between - latency, not work.  Lane group c = lane >> 4 walks chain s0 + c with its own step vectors and
  // operand area in LDS; the same arithmetic per entry and the same order of every sum as chain_step() (bitwise the same steps), a quarter
  // of the sequential steps per wavefront.  On the chain levels node (k, s) = level_node_start[k] + s has the one child edge
  // node_child_start[level_node_start[k]] + s leading to node (k + 1, s) (checked by the runtime when it sets chain_level).
  constexpr int FW4_CH = ((3 * 16 + FW_N + 1) / 2) * 2, FW4_PL = (FW_N + 15) / 16;
  constexpr bool FW4 = (DOMPC_FW4 != 0) && NA <= 16 && NV <= 16 && 4 * FW4_CH <= EL_SIZE;
  if (FW4 && GS == 64) {
    const int S = A.level_node_start[A.N + 1] - A.level_node_start[A.N];
    const int c4 = lane >> 4, ll = lane & 15;
    ldsd* C = Ld + c4 * FW4_CH;
    ldsd *DX = C, *DV = C + 16, *DXN = C + 32, *IN = C + 48;
    struct Ix { int uo, eo, xoc, row0; unsigned ndc; };       // per-lane (= per-chain) indices of a step, requested with its operands
    const int cw = (S + ng - 1) / ng < 4 ? (S + ng - 1) / ng : 4;      // chains per wavefront (one problem alone: every chain has its own wavefront)
    for (int s0 = cw * gid; s0 < S && cl < A.N; s0 += cw * ng) {
      const bool here = c4 < cw && s0 + c4 < S;
      const int sc = here ? s0 + c4 : S - 1;                  // (lane groups without a chain repeat the last one and store nothing)
      const bool on = here && mk_n(A, A.level_node_start[A.N] + sc);
      double v[FW4_PL];
      auto load4 = [&](int k, Ix& ix) {
        const int n = A.level_node_start[k] + sc, e = A.node_child_start[A.level_node_start[k]] + sc, cn = A.level_node_start[k + 1] + sc;
        const unsigned nd0 = (unsigned)n * (unsigned)ND_SIZE, es0 = (unsigned)e * (unsigned)ES_SIZE, nc0 = (unsigned)cn * (unsigned)ND_SIZE;
#pragma unroll
        for (int q = 0; q < FW4_PL; ++q) {
          const int i = ll + 16 * q;
          double x = 0.0;
          if (i < NV * NA) x = ldoff(Q.nd, nd0 + (unsigned)(ND_K + i));
          else if (i < FW_K) x = ldoff(Q.nd, nd0 + (unsigned)(ND_KV + i - NV * NA));
          else if (i < FW_K + NX * NA) x = ldoff(Q.es, es0 + (unsigned)(ES_AB + i - FW_K));
          else if (i < FW_K + FW_AB) x = ldoff(Q.es, es0 + (unsigned)(ES_CV + i - FW_K - NX * NA));
          else if (i < FW_K + FW_AB + NX * NA) x = ldoff(Q.nd, nc0 + (unsigned)(ND_P + i - FW_K - FW_AB));
          else if (i < FW_N) x = ldoff(Q.nd, nc0 + (unsigned)(ND_PV + i - FW_K - FW_AB - NX * NA));
          v[q] = x;
        }
        ix.uo = A.node_u_off[n];
        ix.eo = NS > 0 ? A.node_eps_off[n] : 0;
        ix.xoc = A.node_x_off[cn];
        ix.row0 = A.edge_row0[e];
        ix.ndc = nc0;
      };
      Ix cur, nxt;
      load4(cl, cur);
      if (ll < NA) DX[ll] = ldoff(Q.nd, (unsigned)(A.level_node_start[cl] + sc) * (unsigned)ND_SIZE + (unsigned)(ND_DXT + ll));
      for (int k = cl; k < A.N; ++k) {
#pragma unroll
        for (int q = 0; q < FW4_PL; ++q) {
          const int i = ll + 16 * q;
          if (i < FW_N) IN[i] = v[q];
        }
        if (k + 1 < A.N) load4(k + 1, nxt);                  // (in flight during the step)
        T.gsync();
        const ldsd *K_ = IN, *KV_ = K_ + NV * NA, *AB_ = IN + FW_K, *CV_ = AB_ + NX * NA, *PC_ = IN + FW_K + FW_AB, *PV_ = PC_ + NX * NA;
        if (ll < NV) {
          double t = KV_[ll];
#pragma unroll
          for (int a = 0; a < NA; ++a) t += K_[ll * NA + a] * DX[a];
          DV[ll] = t;
          if (on) {
            if (ll < NU) Q.dx[cur.uo + ll] = t;
            else Q.dx[cur.eo + ll - NU] = t;
          }
        }
        T.gsync();
        if (ll < NA) {
          double t;
          if (ll < NX) {
            t = CV_[ll];
#pragma unroll
            for (int b = 0; b < NX; ++b) t += AB_[ll * NA + b] * DX[b];
#pragma unroll
            for (int b = 0; b < NU; ++b) t += AB_[ll * NA + NX + b] * DV[b];
            if (on) Q.dx[cur.xoc + ll] = t;
          } else {
            t = DV[ll - NX];
          }
          if (on) Q.nd[cur.ndc + (unsigned)(ND_DXT + ll)] = t;
          DXN[ll] = t;
        }
        T.gsync();
        if (ll < NX) {
          double t = PV_[ll];
#pragma unroll
          for (int b = 0; b < NA; ++b) t += PC_[ll * NA + b] * DXN[b];
          if (on) Q.dlam[cur.row0 + NW + ll] = t;
        }
        if (ll < NA) DX[ll] = DXN[ll];
        T.gsync();
        cur = nxt;
      }
    }
    T.sync();
  } else
#endif
  {
    const int S = A.level_node_start[A.N + 1] - A.level_node_start[A.N];
    for (int s_ = gid; s_ < S; s_ += ng) {
      if (!mk_n(A, A.level_node_start[A.N] + s_)) continue;
      if (cl >= A.N) continue;
      double vin[FW_PL];
      load_step(A.level_node_start[cl] + s_, vin);
      for (int a = lane; a < NA; a += GS) Ld[RF_DX + a] = Q.ND(A.level_node_start[cl] + s_)[ND_DXT + a];
      T.gsync();
      for (int k = cl; k < A.N; ++k) {
        double vnx[FW_PL];
        if (k + 1 < A.N) load_step(A.level_node_start[k + 1] + s_, vnx);
        chain_step(A.level_node_start[k] + s_, vin);
        if (k + 1 < A.N) {
#pragma unroll
          for (int q = 0; q < FW_PL; ++q) vin[q] = vnx[q];
        }
      }
    }
    T.sync();
  }
  DOMPC_PF(16)
  // initial-condition multiplier step
  for (int a = T.tid; a < NX; a += T.nt) {
    const double* Nd = Q.ND(0);
    double t = Nd[ND_PV + a];
    for (int b = 0; b < NA; ++b) t += Nd[ND_P + a * NA + b] * Nd[ND_DXT + b];
    Q.dlam[a] = FREE_ROOT ? 0.0 : -t;
  }
  // per edge: dw, d nu, d lambda, nl_cons steps.  The collocation steps come from the stored inverse block,
  //     dw = -G_w^-1 (G_y dy + r),   G_y dy: -C_0j dx / -D_0 dx on the rows of the first element, J_u du on the collocation rows,
  // (W = -G_w^-1 G_y itself is not kept beyond the sweep), the multiplier steps from its transpose.  Everything a lane needs
  // from the per-edge record (its row AND its column of the stored inverse block, Sigma_w, r_w), from the model-output
  // record (its row of H_ww / H_wu, its entries of J_u - read from the dense image, mo_expand) and from the node steps is
  // loaded in ONE batch at the top of the edge.
#ifndef DOMPC_HOST_EMU
  auto stage_fw = [&](int e) {               // LDS-DMA: 64 lanes x 16 B per instruction (see stage_mo)
    const double* ew_ = Q.ew + (int64_t)e * EW_SIZE;
    const double* mo_ = Q.MO(e);
#pragma unroll
    for (int q = 0; q < EW_STAGE / 128; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ew_ + 128 * q + 2 * lane),
                                       (__attribute__((address_space(3))) void*)(Ld + RF_EW + 128 * q), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < MOC_STAGE / 128; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(mo_ + 128 * q + 2 * lane),
                                       (__attribute__((address_space(3))) void*)(Ld + RF_MOC + 128 * q), 16, 0, 0);
  };
#endif
  // ---- adjoint recovery of the continuity multipliers (round 5, DOMPC_ADJ_REFINE).  The chain walk forms the step of the multipliers
  // of a node's incoming continuity rows as d nu = P dx + p.  Near the solution P carries the Sigma entries of active bounds further
  // down the chain (1e9 ... 1e11) in rank-one terms a a' whose contribution a (a' dx) is tiny in exact arithmetic: a' dx is a sum of
  // terms of size 1e-3 that cancel to 1e-11 and keeps an absolute error of 1e-19, times 2e11 = 2e-8 - the floor of the dual
  // residual (measured on member 2048 of the bench batch: the x rows of the linear system are left with 1.9e-7 where
  // the u rows and the rows of the collocation unknowns have 1e-10 ... 1e-13).  The x row of the Newton system of node c itself has
  // no such terms: with every other step known it determines d nu_c,
  //     d nu_c = rx_c + (Sigma_x + delta) dx_c + sum over the child edges e' of c [ G_y' dlambda_w + (omega H_l + H_nl) dy + Jd' dyd ]_x  (+ omega H_m dx_c at a leaf),
  // and G_y has only the collocation coefficients in its x columns (-C_0j, -D_0).  The edges are processed from the last stage
  // upwards (all child edges of a node before its incoming edge); the shares are kept in the p slot of the node records, which
  // nobody reads after the chain walk (first the node's own terms, then - once its incoming edge has used them - that edge's share
  // for the parent: one writer per slot, sums in the order of the children, the same bits in every launch shape).
  // Measured (B = 16 384, 12 members against oracle solves): every member stops in the oracle's iteration (without: 5 of 12 one to
  // four iterations later), mean iteration count 56.574 -> 56.317, kernel time + 1.0 % (this instantiation has no two-edge path).
  constexpr bool adj = ADJ;            // (decided by the caller: forward_adjoint())
  if (adj) {
    for (int it = T.tid; it < A.n_nodes * NX; it += T.nt) {
      const int n = it / NX, a = it % NX, g = A.node_x_off[n] + a;
      const double xv = Q.x[g], l = Q.lb[g], u = Q.ub[g];
      double t = Q.rd[g] + Q.zl[g] - Q.zu[g] + bar_grad(xv, l, u, mu) + (sigma_of(xv, l, u, Q.zl[g], Q.zu[g]) + delta) * Q.dx[g];
      if (A.node_child_count[n] == 0) {
        const double* S_ = Q.ES(A.node_in_edge[n]);
        for (int b = 0; b < NX; ++b) t += S_[ES_MH + a * NX + b] * Q.dx[A.node_x_off[n] + b];
      }
      Q.ND(n)[ND_PV + a] = t;
    }
    T.sync();
  }
#ifndef DOMPC_HOST_EMU
#ifndef DOMPC_FE2
#define DOMPC_FE2 1                 // per-edge part of the forward pass: two edges per wavefront (0: one)
#endif
  // Two edges per wavefront.  The per-edge part keeps NW (<= 32) lanes busy - one row of the edge's block each - and is a sequence of
  // memory round trips and dependent sums like the chain walk above; lanes 0-31 now handle edge 2 p, lanes 32-63 edge 2 p + 1 of a pair,
  // each half with its own step vectors and staging buffer in LDS (the same arithmetic per row and the same order of every sum).
  // Two dense images of the model-output record do not fit the region: a lane's sixteen entries of the record (its row of H_ww | H_wu,
  // its entries of J_u) are read straight from the staged COMPACT record through a table of their positions, built once per pass
  // (position in the compact record, or in a small pool of the model's constants kept in the slack of the staging buffer).
  constexpr int FE_HV = 128, FE_DY = 0, FE_DNU = 16, FE_G = 32, FE_DW = 64, FE_RHS = 96;      // step vectors of a half
  constexpr int FE_SS = EW_STAGE + MOC_STAGE, FE_STG = 2 * FE_HV, FE_POOL = EW_STAGE + MOC_SIZE, FE_TAB = FE_STG + 2 * FE_SS;
  constexpr bool FE2 = (DOMPC_FE2 != 0) && MO_LDS && M > 0 && NI == 1 && DEG > 0 && !DENSE_EDGE && DOMPC_SHARD == 0 && NW <= 32 && NA <= 16 &&
                       NA + NU <= 16 && NE <= 32 && (MOC_STAGE - MOC_SIZE >= 1 + DOMPC_DYN_NC) && (FE_TAB + 128 <= EL_SIZE) &&
                       (PT_STRIDE <= 2 * FE_SS) && LU_N < NW;
  if (FE2 && GS == 64 && !adj) {
    typedef __attribute__((address_space(3))) unsigned short ldsu16_;
    const int h = lane >> 5, l32 = lane & 31;
    ldsd* Lv = Ld + h * FE_HV;
    ldsd* Ls = Ld + FE_STG + h * FE_SS;
    ldsu16_* tab = (ldsu16_*)(Ld + FE_TAB);
    constexpr int NVD = DOMPC_DYN_NV > 0 ? DOMPC_DYN_NV : 1, NCD = DOMPC_DYN_NC > 0 ? DOMPC_DYN_NC : 1;
    {
      // table of this row's entries: position of dense entry d of a point record = compact index (variable entry), pool (constant), zero
      ldsu16_* inv = (ldsu16_*)(Ld + FE_STG);
      for (int d = lane; d < PT_STRIDE; d += 64) inv[d] = 0xffffu;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int v = lane; v < DOMPC_DYN_NV; v += 64) inv[DOMPC_DYN_VIDX[v % NVD]] = (unsigned short)v;
      for (int c = lane; c < DOMPC_DYN_NC; c += 64) inv[DOMPC_DYN_CIDX[c % NCD]] = (unsigned short)(0x8000u | (unsigned)c);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      unsigned ent[16];
      {
        const int rc = l32 < NW ? l32 : 0;
        const int pt = point_of_slot(rc / NX);
#pragma unroll
        for (int b = 0; b < 16; ++b) {
          const int d = (b < NA) ? MOH_H0 + symi(rc % NX, b, NA) : NX + (rc % NX) * NA + NX + (b - NA < NU ? b - NA : 0);
          const unsigned t = inv[d];
          unsigned en = (unsigned)FE_POOL;                                   // 0.0
          if (pt >= 0 && b < NA + NU && t != 0xffffu)
            en = (t & 0x8000u) ? (unsigned)(FE_POOL + 1) + (t & 0x7fffu) : (unsigned)(EW_STAGE + pt * DOMPC_DYN_NV) + t;
          ent[b] = en;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (h == 0) {
#pragma unroll
        for (int b = 0; b < 16; ++b) tab[l32 * 16 + b] = (unsigned short)ent[b];
      }
      if (l32 <= DOMPC_DYN_NC) Ls[FE_POOL + l32] = (l32 == 0) ? 0.0 : DOMPC_DYN_CVAL[(l32 - 1) % NCD];      // (both halves: own pool)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // uniform data of the two edges of a pair, selected per half
    struct EdgeU { int n, cn, row0, woff, uo; bool chain; };
    auto edge_u = [&](int e) {
      EdgeU u;
#if DOMPC_EDGE_PACK
      const auto* ep = A.edge_pack + e * EP_N;
      u.n = ep[EP_PARENT]; u.cn = ep[EP_CHILD]; u.row0 = ep[EP_ROW0]; u.woff = ep[EP_WOFF];
      u.uo = ep[EP_UOFF_PARENT]; u.chain = ep[EP_LEVEL] >= cl;
#else
      u.n = A.edge_parent[e]; u.cn = A.edge_child[e]; u.row0 = A.edge_row0[e]; u.woff = A.edge_w_off[e];
      u.uo = A.node_u_off[u.n]; u.chain = A.edge_level[e] >= cl;
#endif
      return u;
    };
    auto stage2 = [&](int ea, int eb) {          // both edges of a pair: forward record + compact model-output record (exact size: the pool stays)
      const int es[2] = {ea, eb};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const double* ew_ = Q.ew + (int64_t)es[k] * EW_SIZE;
        const double* mo_ = Q.MO(es[k]);
        ldsd* dst = Ld + FE_STG + k * FE_SS;
#pragma unroll
        for (int q = 0; q < EW_STAGE / 128; ++q)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ew_ + 128 * q + 2 * lane),
                                           (__attribute__((address_space(3))) void*)(dst + 128 * q), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < (MOC_SIZE + 127) / 128; ++q)
          if (128 * q + 2 * lane < MOC_SIZE)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(mo_ + 128 * q + 2 * lane),
                                             (__attribute__((address_space(3))) void*)(dst + EW_STAGE + 128 * q), 16, 0, 0);
      }
    };
    auto load_dy2 = [&](const EdgeU& u, double& dy_, double& dnu_, double& cr_) {
      const int a0 = l32 < NA ? l32 : 0;
      dy_ = (a0 < NX) ? Q.ND(u.n)[ND_DXT + a0] : Q.dx[u.uo + a0 - NX];
      dnu_ = Q.dlam[u.row0 + NW + (l32 < NX ? l32 : 0)];
      cr_ = Q.c[u.row0 + (l32 < NW ? l32 : 0)];
    };
    auto pick = [&](const EdgeU& a, const EdgeU& b) {
      EdgeU u;
      u.n = h ? b.n : a.n; u.cn = h ? b.cn : a.cn; u.row0 = h ? b.row0 : a.row0; u.woff = h ? b.woff : a.woff;
      u.uo = h ? b.uo : a.uo; u.chain = h ? b.chain : a.chain;
      return u;
    };
    const int np = (A.n_edges + 1) / 2;
    bool staged = false;
    double dy0 = 0.0, dnu0 = 0.0, cr0 = 0.0;
    for (int p_ = gid; p_ < np; p_ += ng) {
      const int ea = 2 * p_, eb = (2 * p_ + 1 < A.n_edges) ? 2 * p_ + 1 : 2 * p_;
      const bool on = (h == 0) || (2 * p_ + 1 < A.n_edges);      // (odd number of edges: the second half of the last pair repeats the edge and stores nothing)
      const EdgeU U = pick(edge_u(ea), edge_u(eb));
      const int e = h ? eb : ea;
      const double* Nc = Q.ND(U.cn);
      const int row0 = U.row0;
      constexpr int LU1 = LU_N > 0 ? LU_N : 1;
      constexpr int NU1 = NU > 0 ? NU : 1;
      constexpr int ELR = (DEG + 1) * NX > 0 ? (DEG + 1) * NX : 1;
      double hrow[NA], ju[NU1], rw_r, sg_r, inv_c[LU1], inv_r[LU1], c_r;
      if (!staged) { stage2(ea, eb); load_dy2(U, dy0, dnu0, cr0); staged = true; }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      {
        const int r = l32, rc = r < NW ? r : 0;
        rw_r = Ls[EW_RW + rc];
        sg_r = Ls[EW_SIGW + rc];
        const ldsu16_* tr = tab + l32 * 16;
#pragma unroll
        for (int b = 0; b < NA; ++b) hrow[b] = Ls[tr[b]];
#pragma unroll
        for (int u = 0; u < NU; ++u) ju[u] = Ls[tr[NA + u]];
        const int rl = r < LU_N ? r : 0;
#pragma unroll
        for (int k2 = 0; k2 < LU_N; ++k2) {
          inv_c[k2] = Ls[EW_LU + k2 * LU_N + rl];
          inv_r[k2] = Ls[EW_LU + rl * LU_N + k2];
        }
      }
      double dy_n = 0.0, dnu_n = 0.0, cr_n = 0.0;
      {
        // everything of this pair is in registers: hand the staging buffers to the next pair of this wavefront
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int pn = p_ + ng;
        if (pn < np) {
          const int na = 2 * pn, nb = (2 * pn + 1 < A.n_edges) ? 2 * pn + 1 : 2 * pn;
          stage2(na, nb);
          load_dy2(pick(edge_u(na), edge_u(nb)), dy_n, dnu_n, cr_n);
        }
      }
      if (l32 < NA) Lv[FE_DY + l32] = dy0;
      if (U.chain && l32 < NX) Lv[FE_DNU + l32] = dnu0;
      c_r = cr0;
      if (!U.chain && l32 < NX) {
        double t = Nc[ND_PV + l32];
#pragma unroll
        for (int b = 0; b < NA; ++b) t += Nc[ND_P + l32 * NA + b] * Nc[ND_DXT + b];
        Lv[FE_DNU + l32] = t;
        if (on) Q.dlam[row0 + NW + l32] = t;
      }
      DOMPC_PF(17)
      T.gsync();
      {
        const int r = l32;
        // g = G_y dy + r on the rows of the stored block
        if (r < LU_N) {
          const int i = r / ELR, rr = r % ELR, jj = rr / NX, a = rr % NX;
          double t = c_r;
          if (jj < DEG) {
#pragma unroll
            for (int u = 0; u < NU; ++u) t += ju[u] * Lv[FE_DY + NX + u];
            if (i == 0) t -= tab_sel(DOMPC_C, jj + 1, DEG > 0 ? 1 : 0, DEG > 0 ? DEG : 1) * Lv[FE_DY + a];
          } else if (i == 0) {
            t -= DOMPC_D[0] * Lv[FE_DY + a];
          }
          Lv[FE_G + r] = t;
        }
        T.gsync();
        // dw = -G_w^-1 g: the rows of the stored block ...
        if (r < LU_N) {
          double t = 0.0;
#pragma unroll
          for (int k2 = 0; k2 < LU_N; ++k2) t -= inv_r[k2] * Lv[FE_G + k2];
          Lv[FE_DW + r] = t;
          if (on) Q.dx[U.woff + r] = t;
        }
        // ... and the end-point rows from the continuity equation  dw_e = sum_s D_s dw_s + D_0 dx - r_e
        T.gsync();
        if (r >= LU_N && r < NW) {
          const int a = r - LU_N;
          double t = DOMPC_D[0] * Lv[FE_DY + a] - c_r;
#pragma unroll
          for (int s_ = 1; s_ <= DEG; ++s_) t += DOMPC_D[s_] * Lv[FE_DW + (s_ - 1) * NX + a];
          Lv[FE_DW + r] = t;
          if (on) Q.dx[U.woff + r] = t;
        }
        T.gsync();
        DOMPC_PF(18)
        // rhs = -(rw + (Sigma_w+delta) dw + Hww dw + Hwu du + S' dnu)
        if (r < NW) {
          const int sl = r / NX;
          double t = rw_r + sg_r * Lv[FE_DW + r];
          if (r >= (M - 1) * NX) t += Lv[FE_DNU + r - (M - 1) * NX];
#pragma unroll
          for (int b = 0; b < NX; ++b) t += hrow[b] * Lv[FE_DW + sl * NX + b];
#pragma unroll
          for (int b = 0; b < NU; ++b) t += hrow[NX + b] * Lv[FE_DY + NX + b];
          Lv[FE_RHS + r] = -t;
        }
        T.gsync();
        DOMPC_PF(19)
        // d lambda = G_w^-T rhs   (G_w^-T = [[Gi', -Gi'E'], [0, I]])
        if (r < NW) {
          double t = 0.0;
          if (r < LU_N) {
#pragma unroll
            for (int k2 = 0; k2 < LU_N; ++k2)
              t += inv_c[k2] * (Lv[FE_RHS + k2] + DOMPC_D[k2 / NX + 1] * Lv[FE_RHS + LU_N + k2 % NX]);
          } else {
            t = Lv[FE_RHS + r];
          }
          if (on) Q.dlam[row0 + r] = t;
        }
      }
      if (NE > 0) {
        const double* S_ = Q.ES(e);
        if (l32 < NE && on) {
          const int i = l32;
          double t = S_[ES_RDN + i];
          for (int b = 0; b < NA; ++b) t += Q.EW(e, EW_JD + i * NA + b) * Lv[FE_DY + b];
          if (!EPS_GLOBAL && nl_slack(i) >= 0) t -= Q.sgn[e * NE1 + i] * Q.dx[A.node_eps_off[U.n] + nl_slack(i)];
          Q.ds[e * NE1 + i] = t;
          Q.dlam[row0 + NW + NX + i] = (S_[ES_SIGS + i] + delta) * t + S_[ES_RSN + i];
        }
      }
      T.gsync();
      dy0 = dy_n; dnu0 = dnu_n; cr0 = cr_n;
      DOMPC_PF(20)
    }
  } else
#endif
  {
  const MocMap mm = moc_map(lane, GS);
  if (MO_COMPACT && M > 0) mo_image_init(Ld + RF_IMG, lane, GS);
  int fw_staged = -1;                        // edge whose records are in (on their way into) the staging area
  double dy0 = 0.0, dnu0 = 0.0, cr0 = 0.0;   // this lane's entry of dy / d nu / the residual of the edge, requested one edge ahead
  bool have_pre = false;
  auto load_dy = [&](int e, double& dy_, double& dnu_, double& cr_) {
    const int n = A.edge_parent[e];
    const int a0 = lane < NA ? lane : 0;
    dy_ = (a0 < NX) ? Q.ND(n)[ND_DXT + a0] : Q.dx[A.node_u_off[n] + a0 - NX];
    // (adjoint recovery: the own terms of the child node's x rows in place of the chain walk's d nu)
    dnu_ = adj ? Q.ND(A.edge_child[e])[ND_PV + (lane < NX ? lane : 0)] : Q.dlam[A.edge_row0[e] + NW + (lane < NX ? lane : 0)];
    cr_ = Q.c[A.edge_row0[e] + (lane < NW ? lane : 0)];
  };
  (void)fw_staged; (void)have_pre; (void)cr0;
  // adjoint recovery, chain levels: the share of the edge just processed for its parent node stays in registers - the next edge of the
  // chain is that node's incoming edge (no trip through memory on the serial path)
  constexpr int NXPL = (NX + GS_C - 1) / GS_C > 0 ? (NX + GS_C - 1) / GS_C : 1;
  double carry[NXPL];
  int carry_node = -1;
  (void)carry; (void)carry_node;
  // Order of the edges.  Without the adjoint recovery the edges are independent: group g takes e = g, g + ng, ...  With it every edge comes
  // after the child edges of its child node: segment 0 - each group walks its scenario chains from the last stage up to the first chain
  // level (no barrier: one wavefront owns a chain); segments 1 ... cl - the branching levels from the lowest to the root, the edges of a
  // level over the groups, a barrier after each.  Edge (k, s) of the chain levels = first edge of level k + s (as in the chain walk).
  // (the chain levels have S_ch edges each, numbered level by level: one subtraction per step, no table look-ups on the serial path)
  const int S_ch = A.level_node_start[A.N + 1] - A.level_node_start[A.N];
  auto lvl_e0 = [&](int k) { return k < A.N ? A.node_child_start[A.level_node_start[k]] : A.n_edges; };
  const int nseg = adj ? cl + 1 : 1;
  const int e_cl = adj ? lvl_e0(cl) : 0, e_bot = e_cl + (A.N - 1 - cl) * S_ch;      // first edge of the first / the last chain level
  for (int seg = 0; seg < nseg; ++seg) {
  const int e_lo = (adj && seg > 0) ? lvl_e0(cl - seg) : 0, e_hi = (adj && seg > 0) ? lvl_e0(cl - seg + 1) : A.n_edges;
  auto seq_first = [&]() -> int {
    if (!adj) return gid < A.n_edges ? gid : -1;
    if (seg == 0) return (cl < A.N && gid < S_ch) ? e_bot + gid : -1;
    return e_lo + gid < e_hi ? e_lo + gid : -1;
  };
  auto seq_next = [&](int e) -> int {
    if (adj && seg == 0) {
      if (e - S_ch >= e_cl) return e - S_ch;
      return e - e_cl + ng < S_ch ? e_bot + (e - e_cl) + ng : -1;
    }
    return e + ng < e_hi ? e + ng : -1;
  };
  for (int e = seq_first(), e_nx = -1; e >= 0; e = e_nx) {
    e_nx = seq_next(e);
    if (!mk_e(A, e)) continue;
    const int n = A.edge_parent[e], cn = A.edge_child[e];
    const double* Nd = Q.ND(n);
    const double* Nc = Q.ND(cn);
    const int row0 = A.edge_row0[e];
    const bool chain_edge = A.edge_level[e] >= cl;          // its d nu was formed by the chain walk
    if constexpr (DENSE_EDGE) {
      // DAE model / rows on the edge unknowns: dense path (dompc_dae.h) - dy of the parent node and d nu of the end-point rows staged, then the edge
      for (int a = lane; a < NA; a += GS) Ld[dae::DF_DY + a] = (a < NX) ? Nd[ND_DXT + a] : Q.dx[A.node_u_off[n] + a - NX];
      for (int a = lane; a < NX; a += GS) {
        double t;
        if (chain_edge) t = Q.dlam[row0 + NW + a];
        else {
          t = Nc[ND_PV + a];
          for (int b = 0; b < NA; ++b) t += Nc[ND_P + a * NA + b] * Nc[ND_DXT + b];
          Q.dlam[row0 + NW + a] = t;
        }
        Ld[dae::DF_DNU + a] = t;
      }
      T.gsync();
      forward_edge_dae(T, Q, e, delta, lane, GS, Ld);
      continue;
    }
    constexpr int RPL = (NW1 + GS_C - 1) / GS_C;          // rows (= columns of G_w^-1) per lane: 1 on the device
    constexpr int LU1 = LU_N > 0 ? LU_N : 1;
    constexpr int NU1 = NU > 0 ? NU : 1;
    constexpr int ELR = (DEG + 1) * NX > 0 ? (DEG + 1) * NX : 1;      // rows of one finite element
    double hrow[RPL][NA], ju[RPL][NU1], rw_r[RPL], sg_r[RPL], inv_c[RPL][LU1], inv_r[RPL][LU1], c_r[RPL];
#ifndef DOMPC_HOST_EMU
    if (MO_LDS && fw_staged != e) { stage_fw(e); fw_staged = e; }
    if (GS > 1 && !have_pre) load_dy(e, dy0, dnu0, cr0);
    if (MO_LDS) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      mo_expand(Ld + RF_IMG, (const ldsd*)(Ld + RF_MOC), mm, lane, GS);
    }
#define EWV(i) (MO_LDS ? (double)Ld[RF_EW + (i)] : Q.EW(e, (i)))
#else
    if (MO_COMPACT && M > 0) mo_expand(Ld + RF_IMG, Q.MO(e), mm, lane, GS);
#define EWV(i) Q.EW(e, (i))
#endif
#define MOVF(i) (MO_COMPACT ? (double)Ld[RF_IMG + (i)] : Q.MO(e)[(i)])
    if (M > 0) {
#pragma unroll
      for (int q = 0; q < RPL; ++q) {
        const int r = lane + q * GS;
        const int rc = r < NW ? r : 0;
        const int pt = point_of_slot(rc / NX);
        rw_r[q] = EWV(EW_RW + rc);
        sg_r[q] = EWV(EW_SIGW + rc);
        const int ptc = pt >= 0 ? pt : 0;
#pragma unroll
        for (int b = 0; b < NA; ++b) hrow[q][b] = MOVF(MO_PT + ptc * PT_STRIDE + MOH_H0 + symi(rc % NX, b, NA));
#pragma unroll
        for (int u = 0; u < NU; ++u) ju[q][u] = MOVF(MO_PT + ptc * PT_STRIDE + NX + (rc % NX) * NA + NX + u);
        const int rl = r < LU_N ? r : 0;
#pragma unroll
        for (int k2 = 0; k2 < LU_N; ++k2) {
          inv_c[q][k2] = EWV(EW_LU + k2 * LU_N + rl);     // column r of the stored block (multiplier steps)
          inv_r[q][k2] = EWV(EW_LU + rl * LU_N + k2);     // row r (collocation steps)
        }
        if (pt < 0) {
#pragma unroll
          for (int b = 0; b < NA; ++b) hrow[q][b] = 0.0;
#pragma unroll
          for (int u = 0; u < NU; ++u) ju[q][u] = 0.0;
        }
        c_r[q] = (GS > 1) ? 0.0 : Q.c[row0 + rc];
      }
    }
#undef EWV
#undef MOVF
    // (adjoint recovery: this lane's column of Jd and the weight of the edge, before the staging area changes hands)
    constexpr int NE1_ = NE > 0 ? NE : 1;
    double jd_r[NXPL][NE1_];
    double omh_a = 0.0;
    if (adj) {
      omh_a = A.edge_omega[e] * Q.sf;
#define EWV(i) (MO_LDS ? (double)Ld[RF_EW + (i)] : Q.EW(e, (i)))
#pragma unroll
      for (int q = 0; q < NXPL; ++q) {
        const int a = lane + q * GS;
#pragma unroll
        for (int i = 0; i < NE; ++i) jd_r[q][i] = EWV(EW_JD + i * NA + (a < NX ? a : 0));
      }
#undef EWV
    }
    (void)jd_r; (void)omh_a;
    double dy_n = 0.0, dnu_n = 0.0, cr_n = 0.0;
    bool pre_n = false;
#ifndef DOMPC_HOST_EMU
    if (MO_LDS) {
      // everything of this edge is in registers: hand the staging area to the next edge of this wavefront
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (e_nx >= 0 && mk_e(A, e_nx)) {
        stage_fw(e_nx);
        fw_staged = e_nx;
        if (GS > 1) { load_dy(e_nx, dy_n, dnu_n, cr_n); pre_n = true; }
      }
    }
#endif
    {
      if (GS > 1) {
        if (lane < NA) Ld[RF_DY + lane] = dy0;
        if (!adj && chain_edge && lane < NX) Ld[RF_DNU + lane] = dnu0;
        c_r[0] = cr0;
#pragma unroll
        for (int q = 1; q < RPL; ++q) {              // (more than 64 unknowns per interval, round 5: the rows beyond the first 64 - only entry 0 is requested one edge ahead)
          const int r = lane + q * GS;
          c_r[q] = Q.c[row0 + (r < NW ? r : 0)];
        }
      } else {
        for (int a = 0; a < NA; ++a) Ld[RF_DY + a] = (a < NX) ? Nd[ND_DXT + a] : Q.dx[A.node_u_off[n] + a - NX];
        if (chain_edge)
          for (int a = 0; a < NX; ++a) Ld[RF_DNU + a] = Q.dlam[row0 + NW + a];
      }
    }
    DOMPC_PF(17)
    if (adj) {
      // adjoint recovery: the shares of the child edges of `cn` are complete (they were processed before this edge)
      const int cs_ = A.node_child_start[cn], cc_ = A.node_child_count[cn];
      const bool in_regs = cc_ == 1 && carry_node == cn;
#ifndef DOMPC_HOST_EMU
      if (!in_regs && cc_ > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      T.gsync();
#pragma unroll
      for (int q = 0; q < NXPL; ++q) {
        const int a = lane + q * GS;
        if (a < NX) {
          double t = (GS > 1 && q == 0) ? dnu0 : Nc[ND_PV + a];             // (the node's own terms)
          if (in_regs) t += carry[q];
          else
            for (int j = 0; j < cc_; ++j) t += Q.ND(A.edge_child[cs_ + j])[ND_PV + a];      // + the shares of its child edges, in their order
          Ld[RF_DNU + a] = t;
          Q.dlam[row0 + NW + a] = t;
        }
      }
    } else if (!chain_edge)
      for (int a = lane; a < NX; a += GS) {
        double t = Nc[ND_PV + a];
#pragma unroll
        for (int b = 0; b < NA; ++b) t += Nc[ND_P + a * NA + b] * Nc[ND_DXT + b];
        Ld[RF_DNU + a] = t;
        Q.dlam[row0 + NW + a] = t;
      }
    T.gsync();
    if (M > 0) {
      const int woff = A.edge_w_off[e];
      // g = G_y dy + r on the rows of the stored block
#pragma unroll
      for (int q = 0; q < RPL; ++q) {
        const int r = lane + q * GS;
        if (r < LU_N) {
          const int i = r / ELR, rr = r % ELR, jj = rr / NX, a = rr % NX;
          double t = c_r[q];
          if (jj < DEG) {
#pragma unroll
            for (int u = 0; u < NU; ++u) t += ju[q][u] * Ld[RF_DY + NX + u];
            if (i == 0) t -= ((NI == 1) ? tab_sel(DOMPC_C, jj + 1, DEG > 0 ? 1 : 0, DEG > 0 ? DEG : 1) : DOMPC_C[jj + 1]) * Ld[RF_DY + a];
          } else if (i == 0) {
            t -= DOMPC_D[0] * Ld[RF_DY + a];
          }
          Ld[RF_G + r] = t;
        }
      }
      T.gsync();
      // dw = -G_w^-1 g: the rows of the stored block ...
#pragma unroll
      for (int q = 0; q < RPL; ++q) {
        const int r = lane + q * GS;
        if (r < LU_N) {
          double t = 0.0;
#pragma unroll
          for (int k2 = 0; k2 < LU_N; ++k2) t -= inv_r[q][k2] * Ld[RF_G + k2];
          Ld[RF_DW + r] = t;
          Q.dx[woff + r] = t;
        }
      }
      if (LU_N < NW) {
        // ... and (single finite element: G_w^-1 = [[Gi, 0], [-E Gi, I]]) the end-point rows from the continuity equation
        //     dw_e = sum_s D_s dw_s + D_0 dx - r_e
        T.gsync();
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
          const int r = lane + q * GS;
          if (r >= LU_N && r < NW) {
            const int a = r - LU_N;
            double t = DOMPC_D[0] * Ld[RF_DY + a] - c_r[q];
#pragma unroll
            for (int s_ = 1; s_ <= DEG; ++s_) t += DOMPC_D[s_] * Ld[RF_DW + (s_ - 1) * NX + a];
            Ld[RF_DW + r] = t;
            Q.dx[woff + r] = t;
          }
        }
      }
      T.gsync();
      DOMPC_PF(18)
      // rhs = -(rw + (Sigma_w+delta) dw + Hww dw + Hwu du + S' dnu)
#pragma unroll
      for (int q = 0; q < RPL; ++q) {
        const int r = lane + q * GS;
        if (r < NW) {
          const int sl = r / NX;
          double t = rw_r[q] + sg_r[q] * Ld[RF_DW + r];      // (the stored Sigma_w holds the inertia correction, Prob::dsw = delta)
          if (r >= (M - 1) * NX) t += Ld[RF_DNU + r - (M - 1) * NX];
#pragma unroll
          for (int b = 0; b < NX; ++b) t += hrow[q][b] * Ld[RF_DW + sl * NX + b];
#pragma unroll
          for (int b = 0; b < NU; ++b) t += hrow[q][NX + b] * Ld[RF_DY + NX + b];
          Ld[RF_RHS + r] = -t;
        }
      }
      T.gsync();
      DOMPC_PF(19)
      // d lambda = G_w^-T rhs
#pragma unroll
      for (int q = 0; q < RPL; ++q) {
        const int r = lane + q * GS;
        if (r < NW) {
          double t = 0.0;
          if (LU_N == NW) {
#pragma unroll
            for (int k2 = 0; k2 < LU_N; ++k2) t += inv_c[q][k2] * Ld[RF_RHS + k2];
          } else if (r < LU_N) {
            // G_w^-T = [[Gi', -Gi'E'], [0, I]]: the continuity part of the right-hand side folds into the collocation part
#pragma unroll
            for (int k2 = 0; k2 < LU_N; ++k2)
              t += inv_c[q][k2] * (Ld[RF_RHS + k2] + DOMPC_D[k2 / NX + 1] * Ld[RF_RHS + LU_N + k2 % NX]);
          } else {
            t = Ld[RF_RHS + r];
          }
          Q.dlam[row0 + r] = t;
          if (adj) Ld[RF_G + r] = t;            // (the g vector is dead: d lambda_w for the parent's sum below)
        }
      }
    }
    if (NE > 0) {
      const double* S_ = Q.ES(e);
      for (int i = lane; i < NE; i += GS) {
        double t = S_[ES_RDN + i];
        for (int b = 0; b < NA; ++b) t += Q.EW(e, EW_JD + i * NA + b) * Ld[RF_DY + b];
        if (!EPS_GLOBAL && nl_slack(i) >= 0) t -= Q.sgn[e * NE1 + i] * Q.dx[A.node_eps_off[n] + nl_slack(i)];      // (shared slacks: their step is part of the residual, eps_schur_apply)
        Q.ds[e * NE1 + i] = t;
        Q.dlam[row0 + NW + NX + i] = (S_[ES_SIGS + i] + delta) * t + S_[ES_RSN + i];
        if (adj) Ld[RF_RHS + i] = Q.dlam[row0 + NW + NX + i];      // (d y_d for the parent's sum)
      }
    }
    T.gsync();
    if (adj && M > 0) {
      // this edge's share of the x rows of its parent node: G_y' dlambda_w (x columns: -C_0j on the collocation rows, -D_0 on the
      // continuity rows of the element), the x rows of omega H_l + H_nl times dy, Jd' dyd
#pragma unroll
      for (int q = 0; q < NXPL; ++q) {
        const int a = lane + q * GS;
        if (a < NX) {
          double t = -DOMPC_D[0] * Ld[RF_G + LU_N + a];
#pragma unroll
          for (int j = 1; j <= DEG; ++j) t -= DOMPC_C[0 * (DEG + 1) + j] * Ld[RF_G + (j - 1) * NX + a];
          for (int b = 0; b < NA; ++b) {
            const int ip = symi(a, b, NA);
            double hv = omh_a * (MO_COMPACT ? (double)Ld[RF_IMG + MO_LT + 1 + NA + ip] : Q.MO(e)[MO_LT + 1 + NA + ip]);
            if (NE > 0) hv += MO_COMPACT ? (double)Ld[RF_IMG + MO_NL + NE + NE * NA + ip] : Q.MO(e)[MO_NL + NE + NE * NA + ip];
            t += hv * Ld[RF_DY + b];
          }
#pragma unroll
          for (int i = 0; i < NE; ++i) t += jd_r[q][i] * Ld[RF_RHS + i];
          Q.ND(cn)[ND_PV + a] = t;              // (in the slot of the child node, whose own terms have been used: one writer per slot)
          carry[q] = t;
        }
      }
      carry_node = n;
      T.gsync();
    }
    dy0 = dy_n; dnu0 = dnu_n; cr0 = cr_n; have_pre = pre_n;
    DOMPC_PF(20)
  }
  if (adj) T.sync();          // (the shares of this segment's edges are visible to the groups of the next one)
  }
  }
  if (adj) {
    const int cs_ = A.node_child_start[0], cc_ = A.node_child_count[0];
    for (int a = T.tid; a < NX; a += T.nt) {                                  // initial-condition rows: + lambda in the root's x rows
      double t = Q.ND(0)[ND_PV + a];
      for (int j = 0; j < cc_; ++j) t += Q.ND(A.edge_child[cs_ + j])[ND_PV + a];
      Q.dlam[a] = -t;
    }
  }
  // dummies (variables in no constraint / cost): independent scalar Newton steps
  for (int d = T.tid; d < A.n_dummy; d += T.nt) {
    const int g = A.dummy_idx[d];
    const double sg = sigma_of(Q.x[g], Q.lb[g], Q.ub[g], Q.zl[g], Q.zu[g]) + delta;
    Q.dx[g] = sg > 0.0 ? -bar_grad(Q.x[g], Q.lb[g], Q.ub[g], mu, !(Q.soc & 2)) / sg : 0.0;
  }
  // (the bound multiplier steps dz are functions of (x, bound, z, dx, mu): formed where they are used - dz_lo / dz_up)
  T.sync();
}

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
// dsw: the inertia correction this sweep folds into the condensed blocks; remembered in Q for the Riccati passes
DOMPC_DEV inline int run_sweep(const Thr& T, Prob& Q, int b, int slot, double mu, int soc = 0, double dsw = 0.0) {
  Q.dsw = dsw;
#ifndef DOMPC_HOST_EMU
  if (fine_items(T, *Q.A)) { DOMPC_PHASE_CALL(phase_sweep_fine, mu, dsw, soc) return ufl(r_.rc); }
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
    DOMPC_PHASE_CALL(phase_forward, mu, delta, Q.dsw)
  }
#else
  (void)b; (void)slot;
  if (forward_adjoint(Q, mu)) riccati_forward_t<true>(T, Q, mu, delta);
  else riccati_forward_t<false>(T, Q, mu, delta);
#endif
}


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
  const double n_dual = (double)A.n_g + n_bounds;

  // ---- objective scaling from the gradient at the (pushed) starting point
  double mu = O.mu_init;
  Q.sf = 1.0;
  long long c_sweep = 0, c_bwd = 0, c_fwd = 0, c_ls = 0, c_meas = 0, c_ftb = 0, c_acc = 0, c_t = 0; const long long c_start = prof_clock();
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
    if (mu != mu_before) refresh_mu(T, Q, mu - mu_before);

    // ---- search direction with inertia correction (delta_w on all primal variables)
    delta = 0.0;
    bool first_try = true, dir_ok = true, recs_dirty = false;
    while (true) {
      c_t = prof_clock();
      int fail = (singular0 && delta == 0.0) ? 1 : run_backward(T, Q, b, slot, mu, delta);
      if (EPS_GLOBAL && !fail) {
        run_forward(T, Q, b, slot, mu, delta);            // structured step at c(x), then the Schur complement of the shared slacks
        fail = epsg_build(delta);
        recs_dirty = true;                                // (the vector parts of the records now belong to the last column's residual)
      }
      c_bwd += prof_clock() - c_t;
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
    c_t = prof_clock();
    if (EPS_GLOBAL) { if (epsg_apply(delta, Q.dlam_e)) { if (in_wd) { wd_resume = true; in_wd = false; wd_count = 0; continue; } status = 3; break; } }
    else run_forward(T, Q, b, slot, mu, delta);
    c_fwd += prof_clock() - c_t;

    // ---- fraction to the boundary, directional derivative of the barrier function
    c_t = prof_clock();
    // largest ratios (-dx)/(x - l), dx/(u - x) and (-dz)/z over the bounded variables: the fraction-to-the-boundary steps
    // are tau / ratio (one division at the end instead of one per bound), and the directional derivative of the barrier function
    double r5[5];
    run_step_rules(T, Q, b, slot, mu, r5);
    a_max = (r5[0] > tau) ? tau / r5[0] : 1.0; dphi = r5[2];
    a_z = (r5[1] > tau) ? tau / r5[1] : 1.0;
    c_ftb += prof_clock() - c_t;
    }     // (!wd_resume)
    auto step_rules = [&](double (&r5)[5]) { run_step_rules(T, Q, b, slot, mu, r5); };
    const double theta = E.theta;
    const double phi = E.obj + mu * bar_sum;       // (the barrier sum of the current point was formed when it was a trial point)

    c_t = prof_clock();
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
    c_ls += prof_clock() - c_t;
    // ---- accept the trial point
    c_t = prof_clock();
    const Comp Cp = run_accept(T, Q, b, slot, alpha, a_z, mu);
    if (A.trace && b == 0 && T.tid == 0 && it < A.trace_cap) {
      double* tr = A.trace + 8 * it;
      tr[0] = it; tr[1] = mu; tr[2] = E0; tr[3] = E.e_p; tr[4] = E.e_d; tr[5] = accepted ? alpha : -alpha;
      tr[6] = delta; tr[7] = E.obj / Q.sf;
    }
    if (T.tid == 0) T.fset(6, abort_requested(A));
    T.sync();
    c_acc += prof_clock() - c_t;
    ++it;
    c_t = prof_clock(); bad = run_sweep(T, Q, b, slot, mu, 0, singular0 ? delta_after(delta_last) : 0.0); c_sweep += prof_clock() - c_t;
    ++n_sweeps;
    if (EPS_GLOBAL) epsg_grad(T, Q);
    if (KAPPA_D != 0.0) Q.mu = mu;
    c_t = prof_clock(); E = measure(T, Q, &Cp); c_meas += prof_clock() - c_t;
  }

  // ---- outputs (unscaled multipliers, CasADi sign convention)
  if (A.trace && b == 0 && T.tid == 0 && A.trace_cap > 8) { double* tr = A.trace + 8 * (A.trace_cap - 1); tr[0] = (double)c_sweep; tr[1] = (double)c_bwd; tr[2] = (double)c_fwd; tr[3] = (double)c_ls; tr[4] = (double)c_meas; tr[5] = (double)(prof_clock() - c_start); tr[6] = (double)c_ftb; tr[7] = (double)c_acc; if (T.prof) { double* t2 = A.trace + 8 * (A.trace_cap - 2); for (int i = 0; i < 8; ++i) t2[i] = (double)T.prof[i]; double* t3 = A.trace + 8 * (A.trace_cap - 3); for (int i = 0; i < 8; ++i) t3[i] = (double)T.prof[8 + i]; double* t4 = A.trace + 8 * (A.trace_cap - 4); for (int i = 0; i < 8; ++i) t4[i] = (double)T.prof[16 + i]; if (A.trace_cap > 12) { double* t5 = A.trace + 8 * (A.trace_cap - 5); for (int i = 0; i < 8; ++i) t5[i] = (double)T.prof[24 + i]; } } }
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

}  // namespace dompc
