// dompc_factor.h - structured interior-point solver, part of dompc_kernel.h (included there, inside namespace dompc, in this order:
// dompc_edge.h, dompc_factor.h, dompc_node.h, dompc_riccati.h, dompc_forward.h, dompc_sweep.h, dompc_phases.h, dompc_driver.h).
// Contents: blocked Gauss-Jordan of the collocation block on the FP64 matrix cores; factorisation part of a single-finite-element edge (phase_edge_factor); the per-edge part of the sweep on the fast path.
// Sizes, record layouts, the thread context `Thr`, reductions and the small dense products are in dompc_kernel.h.

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
  // Several finite elements per interval (round 6): every operand the phases of this edge read from global memory is requested in ONE
  // batch before the first store of the edge.  On gfx9 stores count in vmcnt like loads, so a load behind a store waits for the store's
  // round trip (3 - 4 k cycles): the residual stores inside the assembly loop, the per-variable loads of phase 3 behind them, the stage-cost
  // loads of the condensing and of phase 7 behind the record stores cost batch_reactor five such waits per edge.
  constexpr bool GP = (NI > 1 && M > 0);
  constexpr int G_ROWS = NI * (DEG + 1) * NX;
  constexpr int G_RWL = GP ? (G_ROWS + GS_C - 1) / GS_C : 1, G_WPL = GP ? (NW + GS_C - 1) / GS_C : 1, G_APL = GP ? (NA + GS_C - 1) / GS_C : 1;
  constexpr int G_XPL = GP ? (NX + GS_C - 1) / GS_C : 1, G_MHL = GP ? (NX * NX + GS_C - 1) / GS_C : 1, G_CVL = GP ? (NX * (NA + 1) + GS_C - 1) / GS_C : 1;
  constexpr int G_QPL = GP ? (NA * NA + GS_C - 1) / GS_C : 1;
  double gp_res[G_RWL], gp_x[G_WPL], gp_l[G_WPL], gp_u[G_WPL], gp_zl[G_WPL], gp_zu[G_WPL], gp_ce[G_XPL], gp_cv[G_CVL];
  double gp_ltg[G_APL], gp_mg[G_XPL], gp_mh[G_MHL], gp_qlt[G_QPL], gp_qnl[G_QPL], gp_lt0 = 0.0, gp_mt0 = 0.0;
  (void)gp_res; (void)gp_x; (void)gp_l; (void)gp_u; (void)gp_zl; (void)gp_zu; (void)gp_ce; (void)gp_cv;
  (void)gp_ltg; (void)gp_mg; (void)gp_mh; (void)gp_qlt; (void)gp_qnl; (void)gp_lt0; (void)gp_mt0;
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
    // ---- the batch of global loads of the edge (see GP above)
    if (act) {
#pragma unroll
      for (int q = 0; q < G_WPL; ++q) {
        const int col = lane + q * GS, gi = woff + (col < NW ? col : 0);
        gp_x[q] = Q.x[gi]; gp_l[q] = Q.lb[gi]; gp_u[q] = Q.ub[gi]; gp_zl[q] = Q.zl[gi]; gp_zu[q] = Q.zu[gi];
      }
#pragma unroll
      for (int q = 0; q < G_XPL; ++q) {
        const int a = lane + q * GS, ac = a < NX ? a : 0;
        gp_ce[q] = Q.soc ? Q.c[row0 + NW + ac] : w[(M - 1) * NX + ac] - xc[ac];
        gp_mg[q] = last_stage ? mo[MO_MT + 1 + ac] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < G_CVL; ++q) {
        const int it = lane + q * GS, a = (it < NX * (NA + 1) ? it : 0) / (NA + 1);
        gp_cv[q] = Q.soc ? Q.c[row0 + NW + a] : w[(M - 1) * NX + a] - xc[a];
      }
#pragma unroll
      for (int q = 0; q < G_APL; ++q) {
        const int a = lane + q * GS;
        gp_ltg[q] = mo[MO_LT + 1 + (a < NA ? a : 0)];
      }
#pragma unroll
      for (int q = 0; q < G_MHL; ++q) {
        const int a = lane + q * GS, ac = a < NX * NX ? a : 0;
        gp_mh[q] = last_stage ? mo[MO_MT + 1 + NX + symi(ac / NX, ac % NX, NX)] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < G_QPL; ++q) {
        const int it = lane + q * GS, itc = it < NA * NA ? it : 0;
        const int ip = symi(itc / NA, itc % NA, NA);
        gp_qlt[q] = mo[MO_LT + 1 + NA + ip];
        gp_qnl[q] = NE > 0 ? mo[MO_NL + NE + NE * NA + ip] : 0.0;
      }
      gp_lt0 = mo[MO_LT];
      gp_mt0 = last_stage ? mo[MO_MT] : 0.0;
    }
    // ---- phase 2: assemble Mx = [G_w | G_y | r_g] and the residual rows; the point Hessians needed by the
    //      condensing phases are staged in LDS with the same batch of global loads
    if (act) {
      for (int it = lane; it < NCOLL * NA * NA; it += GS)
        Ld[EL_HP + it] = mo[MO_PT + (it / (NA * NA)) * PT_STRIDE + NX + NX * NA + symi((it % (NA * NA)) / NA, it % NA, NA)];
#pragma unroll
      for (int q2 = 0; q2 < G_RWL; ++q2) {
        const int it = lane + q2 * GS;
        gp_res[q2] = 0.0;
        if (it >= G_ROWS) continue;
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
          gp_res[q2] = res;
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
          gp_res[q2] = res;
          Mr[NW + NA] = res;
          Mr[ns_ * NX + a] += 1.0;
          for (int r = 0; r <= DEG; ++r) {
            if (i == 0 && r == 0) Mr[NW + a] -= DOMPC_D[0];
            else Mr[slot_of(i, r) * NX + a] -= DOMPC_D[r];
          }
        }
      }
    }
    T.gsync();
    // ---- phase 3: dual-residual pieces that need G_w / G_y (before they are overwritten); then the stores of phases 2 and 3
    if (act) {
#pragma unroll
      for (int q = 0; q < G_WPL; ++q) {
        const int col = lane + q * GS;
        if (col >= NW) continue;
        double t = 0.0;
#pragma unroll 6
        for (int r = 0; r < NW; ++r) t += Ld[EL_MX + r * NC + col] * Ld[EL_T0 + r];
        if (col >= (M - 1) * NX) t += nu_e[col - (M - 1) * NX];
        const double xv = gp_x[q], l = gp_l[q], u = gp_u[q];
        gp_x[q] = t - gp_zl[q] + gp_zu[q];                       // (the dual residual of the variable, stored below)
        Ld[EL_RW + col] = t + bar_grad(xv, l, u, mu, !(Q.soc & 2));
        Ld[EL_BB + col] = bar_grad(xv, l, u, 1.0);
        Ld[EL_SG + col] = sigma_of(xv, l, u, gp_zl[q], gp_zu[q]);
      }
      for (int b = lane; b < NA; b += GS) {
        double t = 0.0;
#pragma unroll 6
        for (int r = 0; r < NW; ++r) t += Ld[EL_MX + r * NC + NW + b] * Ld[EL_T0 + r];
        Ld[EL_RY + b] = t;          // completed in phase 7
      }
#pragma unroll
      for (int q = 0; q < G_WPL; ++q) {
        const int col = lane + q * GS;
        if (col < NW) { Q.gf[woff + col] = 0.0; Q.rd[woff + col] = gp_x[q]; }
      }
      if (!Q.soc) {
#pragma unroll
        for (int q2 = 0; q2 < G_RWL; ++q2) {
          const int it = lane + q2 * GS;
          if (it < G_ROWS) Q.c[row0 + it] = gp_res[q2];
        }
#pragma unroll
        for (int q = 0; q < G_XPL; ++q) {
          const int a = lane + q * GS;
          if (a < NX) Q.c[row0 + NW + a] = gp_ce[q];
        }
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
    // Round 6: the same elimination in REGISTERS when the matrix has at most one column per lane (NC <= 64; batch_reactor with two finite
    // elements: 24 unknowns, 30 columns).  The LDS version below re-reads the pivot column and the lane's own column and re-writes the own
    // column in every step (3 NW LDS operations) and searches the pivot with NW packed keys per step: 3.4 k cycles per step, 59 % of a
    // batch_reactor solve (tools/gpu_profile.py).  Here a lane keeps its column in registers for the whole elimination, the pivot column
    // arrives as uniform values (v_readlane -> scalar operands of the FMAs), the rows above the pivot's group are skipped (G_w is block
    // lower-triangular in the grouping described above: column kk is zero there and stays zero; likewise the rows below the element(s)
    // the column belongs to - checked numerically for three shapes, tools/check_gj_structure.py), and the pivots are taken in the natural
    // order - the diagonal carries the collocation coefficients C_jj resp. the identity of the continuity rows - under the threshold test of
    // the single-element paths, |a_kk| >= GJ_U max |a_rk| over the remaining rows of the group.  A failed test leaves the LDS matrix
    // untouched and the pivoting version below runs instead.
    bool gj_done = false;
#ifndef DOMPC_HOST_EMU
#ifndef DOMPC_REG_GJ
#define DOMPC_REG_GJ 1
#endif
    if constexpr ((DOMPC_REG_GJ != 0) && NC <= 64 && NW <= 40 && NW > NX) {
      if (GS == 64) {
        constexpr int GJ_STEPS = NW - NX;
        constexpr int EL_ROWS = (DEG + 1) * NX;
        const int cc_ = lane < NC ? lane : 0;
        double col[NW1];
#pragma unroll
        for (int r = 0; r < NW; ++r) col[r] = act ? Ld[EL_MX + r * NC + cc_] : 1.0;      // (lanes >= NC: a copy of column 0, not written back)
        int bad = 0;
#pragma unroll
        for (int kk = 0; kk < GJ_STEPS; ++kk) {
          const int pos = kk % EL_ROWS;
          const int grp0 = kk - pos + (pos < DEG * NX ? 0 : DEG * NX);
          const int grp1 = kk - pos + (pos < DEG * NX ? DEG * NX : EL_ROWS);
          // (column kk is also zero BELOW the rows of its own element - a collocation column - resp. of the next element - the start
          //  state of the next element, i.e. the continuity group -: fill-in reaches those rows only through later pivots)
          const int hi_ = pos < DEG * NX ? kk - pos + EL_ROWS : (kk - pos + 2 * EL_ROWS < NW ? kk - pos + 2 * EL_ROWS : NW);
          double f[NW1];
#pragma unroll
          for (int r = 0; r < NW; ++r) f[r] = (r >= grp0 && r < hi_) ? lane_bcast(col[r], kk) : 0.0;
          double m = 0.0;
#pragma unroll
          for (int r = 0; r < NW; ++r) if (r > kk && r < grp1) m = fmax(m, fabs(f[r]));
          const double akk = fabs(f[kk]);
          bad |= (int)!(akk >= DOMPC_GJ_U * m && akk > 1e-300);
          const double pinv = fast_rcp(akk > 1e-300 ? f[kk] : 1.0);
          const bool own = lane == kk;
          const double prow = own ? pinv : col[kk] * pinv;
#pragma unroll
          for (int r = 0; r < NW; ++r) if (r >= grp0 && r < hi_ && r != kk) col[r] = fma(-f[r], prow, own ? 0.0 : col[r]);
          col[kk] = prow;
        }
        if (!act) bad = 0;
        if (__builtin_amdgcn_readfirstlane(bad) == 0) {
          if (act && lane < NC) {
#pragma unroll
            for (int r = 0; r < NW; ++r) Ld[EL_MX + r * NC + lane] = col[r];
          }
          T.gsync();
          gj_done = true;
        }
      }
    }
#endif
    if (!gj_done) {
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
      if constexpr (GP) {                     // (requested with the first batch of loads of the edge)
        qlt[q] = act ? gp_qlt[q < G_QPL ? q : 0] : 0.0;
        qnl[q] = (act && NE > 0) ? gp_qnl[q < G_QPL ? q : 0] : 0.0;
      } else {
        qlt[q] = act ? MOV(MO_LT + 1 + NA + ip) : 0.0;
        qnl[q] = (act && NE > 0) ? MOV(MO_NL + NE + NE * NA + ip) : 0.0;
      }
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
      if constexpr (GP) {
#pragma unroll
        for (int q = 0; q < G_CVL; ++q) {
          const int it = lane + q * GS;
          if (it >= NX * (NA + 1)) continue;
          const int a = it / (NA + 1), b = it % (NA + 1);
          const double v = Ld[EL_MX + ((M - 1) * NX + a) * MX_LD + MX_W + b];
          if (b < NA) S_[ES_AB + a * NA + b] = v;
          else S_[ES_CV + a] = v + gp_cv[q];
        }
      } else
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
    } else if constexpr (GP) {             // (operands in registers since the first load batch of the edge)
#pragma unroll
      for (int q = 0; q < G_APL; ++q) {
        const int a = lane + q * GS;
        if (a < NA) {
          const double grt = RT_CUSTOM ? (double)Ld[EL_RT + 1 + a] : 0.0;       // d rterm / d (x_n, u_n)
          double r = Ld[EL_RY + a] + om * gp_ltg[q] + grt;
          if (NE > 0)
            for (int i = 0; i < NE; ++i) r += mo[MO_NL + NE + i * NA + a] * yd[i] * Q.sgn[e * NE1 + i];
          S_[ES_GFY + a] = om * gp_ltg[q] + grt;
          S_[ES_RY + a] = r;
          S_[ES_QV + a] = Ld[EL_QV + a] + r;
          S_[ES_QVB + a] = Ld[EL_QV + NA + a];
        }
      }
      if (last_stage) {
#pragma unroll
        for (int q = 0; q < G_XPL; ++q) {
          const int a = lane + q * GS;
          if (a < NX) S_[ES_MG + a] = om * gp_mg[q];
        }
#pragma unroll
        for (int q = 0; q < G_MHL; ++q) {
          const int a = lane + q * GS;
          if (a < NX * NX) S_[ES_MH + a] = omh * gp_mh[q];
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
      double obj = PF ? om * pf_lt0 : (GP ? om * gp_lt0 : om * mo[MO_LT]);
      if (k == A.N - 1) obj += PF ? om * pf_mt0 : (GP ? om * gp_mt0 : om * mo[MO_MT]);
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

