// dompc_quad.h - structured interior-point solver, part of dompc_kernel.h (included there, inside namespace dompc, after dompc_factor.h).
// Contents: the edge sweep of the single-finite-element fast path with FOUR EDGES PER WAVEFRONT (round 6).
//
// What it replaces: the wavefront-per-edge path eval_edge_coop / phase_edge_factor (dompc_factor.h), which stays as the fallback of an edge
// whose collocation block fails the pivot test in its natural order, and as the path of every model outside QUAD_EDGE (dompc_edge.h).
// Reference: everything nlp_jac_g / nlp_hess_l / nlp_grad_f and the factorisation of the collocation rows contribute to
// `r = self.S(**kwargs)` (/root/reference/do_mpc/optimizer.py:770) for one control interval (optimizer.py:905-983, _mpc.py:1189-1275).
//
// Why.  Rounds 3 - 5 measured the wavefront-per-edge sweep at 2 190 vector-ALU issue slots per edge (1 404 instructions + 49 FP64 matrix
// instructions of 16 slots) with the SIMD busy 45 - 50 % of the time: a 20 x 20 block on 64 lanes leaves most lanes idle in every dependent
// chain (the 4 x 4 pivot blocks of the blocked elimination are factorised redundantly by all 64 lanes), FP64 matrix instructions have the
// vector ALU's rate on gfx950, and the dense LDS image of the model-output record (5.9 KB) allowed one edge per wavefront only.
// Here 16 lanes own one edge and a wavefront instruction serves four edges:
//   * lane b < NX of an edge owns state b: its collocation unknowns (one register per slot), their bounds, multipliers and residual rows,
//     and COLUMN (s, b) of the collocation block G_cc for every slot s; lanes NX .. NA-1 own the input columns J_u, lane NA the residual
//     column.  The right-hand-side columns of the parent STATE are not eliminated at all: they are -C[0][j] I, so their part of
//     W = -G_cc^-1 G_y is a combination of the columns of the inverse that the owning lane already holds.
//   * the elimination is an unblocked in-place Gauss-Jordan in registers: per pivot the owning lane's column goes to the 16 lanes of its
//     edge with v_mov_b64_dpp row_newbcast (DPP broadcast inside a row of 16 lanes - one instruction, no LDS, no readlane/SGPR detour);
//     natural pivot order with the threshold test of the old path, the old path as fallback.
//   * no dense image: the entries of the compact model-output record are read where they are needed, addressed at COMPILE time (the
//     structure tables of the generated header are constant expressions): the point Hessians enter the condensing as the sparse matrices
//     they are (industrial_poly: 25 of 91 packed entries per point).
//   * condensing in the same layout: lane c owns column c of [W | w0 | -], T = (H_p + Sigma_p) Z_p is lane-local, Z_p' T takes the rows of
//     Z_p from a staged copy in LDS (group-uniform reads, no vector-ALU slots).
// Measured effect: DESIGN.md section 4 / profiles/r06_*.
#if !defined(DOMPC_HOST_EMU) && DOMPC_DEG >= 1 && DOMPC_M >= 1 && DOMPC_NI == 1 && DOMPC_NX + DOMPC_NU + 2 <= 16 && DOMPC_DEG * DOMPC_DEG * DOMPC_NX <= 64      // (array sizes and lane numbers below; the rest of the conditions: QUAD_EDGE, dompc_edge.h)

#define DOMPC_HAVE_QUAD_HELPERS 1      // rbc / sfor / pin / QD_SB exist (dompc_riccati4.h)
extern "C" __device__ double dompc_dpp_f64(double old, double src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) __asm("llvm.amdgcn.update.dpp.f64");
// value of `v` in lane L of this lane's row of 16 lanes (v_mov_b64_dpp row_newbcast:L)
template <int L>
__device__ inline double rbc(double v) {
  static_assert(L >= 0 && L < 16, "lane inside a row of 16");
  return dompc_dpp_f64(0.0, v, 0x150 + L, 0xf, 0xf, true);
}
// the value is computed HERE (an empty asm the optimiser cannot look through): without it the IR-level sinking pass moves whole chains of
// arithmetic next to their first use, hundreds of instructions later, and keeps their operands alive (in scratch) in between
__device__ inline void pin(double& v) { asm volatile("" : "+v"(v)); }
__device__ inline void pin(int& v) { asm volatile("" : "+v"(v)); }
// d += (value of `src` in lane L of the row of 16) * mul in ONE instruction: v_fmac_f64_dpp with the broadcast as the DPP operand.  The compiler
// has the instruction but never folds a v_mov_b64_dpp into it, and an inline-asm DPP read is outside its hazard recogniser: the caller
// guarantees that `src` was not written by one of the two preceding vector instructions (gfx9 DPP hazard: two wait states).
#ifndef DOMPC_QUAD_FMAC_DPP
#define DOMPC_QUAD_FMAC_DPP 1
#endif
template <int L>
__device__ inline void fmac_rbc(double& d, const double& src, double mul) {
#if DOMPC_QUAD_FMAC_DPP
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(src), "v"(mul), "n"(L));
#else
  d = fma(rbc<L>(src), mul, d);
#endif
}
template <class F, int... I>
__device__ inline void sfor_(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
// f(integral_constant<int, i>) for i = 0 .. N-1: loops whose index has to be a constant expression
template <int N, class F>
__device__ inline void sfor(F&& f) { sfor_(f, std::make_integer_sequence<int, (N > 0 ? N : 0)>{}); }

// where a dense entry of the model-output record lives: kind 0 structural zero, 1 model constant (index into the constant table),
// 2 variable (offset inside the compact record of the edge)
struct MoRef { int kind, off; };
constexpr MoRef moref_in(int d, const int* vidx, int nv, const int* cidx, int nc, int base) {
  for (int k = 0; k < nv; ++k) if (vidx[k] == d) return MoRef{2, base + k};
  for (int k = 0; k < nc; ++k) if (cidx[k] == d) return MoRef{1, k};
  return MoRef{0, 0};
}
constexpr MoRef moref_dyn(int p, int d) { return moref_in(d, DOMPC_DYN_VIDX, DOMPC_DYN_NV, DOMPC_DYN_CIDX, DOMPC_DYN_NC, p * DOMPC_DYN_NV); }
constexpr MoRef moref_lt(int d) { return moref_in(d, DOMPC_LT_VIDX, DOMPC_LT_NV, DOMPC_LT_CIDX, DOMPC_LT_NC, MOC_LT); }
constexpr MoRef moref_nl(int d) { return moref_in(d, DOMPC_NL_VIDX, DOMPC_NL_NV, DOMPC_NL_CIDX, DOMPC_NL_NC, MOC_NL); }
constexpr MoRef moref_mt(int d) { return moref_in(d, DOMPC_MT_VIDX, DOMPC_MT_NV, DOMPC_MT_CIDX, DOMPC_MT_NC, MOC_MT); }
// its value: `rec` = the compact record of this lane's edge in LDS (a constant index into a table with a constant initialiser folds to the literal)
template <int KIND, int OFF>
__device__ inline double mo_dyn_val(const ldsd* rec) { if constexpr (KIND == 2) return rec[OFF]; else if constexpr (KIND == 1) return DOMPC_DYN_CVAL[OFF]; else return 0.0; }
template <int KIND, int OFF>
__device__ inline double mo_lt_val(const ldsd* rec) { if constexpr (KIND == 2) return rec[OFF]; else if constexpr (KIND == 1) return DOMPC_LT_CVAL[OFF]; else return 0.0; }
template <int KIND, int OFF>
__device__ inline double mo_nl_val(const ldsd* rec) { if constexpr (KIND == 2) return rec[OFF]; else if constexpr (KIND == 1) return DOMPC_NL_CVAL[OFF]; else return 0.0; }
template <int KIND, int OFF>
__device__ inline double mo_mt_val(const ldsd* rec) { if constexpr (KIND == 2) return rec[OFF]; else if constexpr (KIND == 1) return DOMPC_MT_CVAL[OFF]; else return 0.0; }
// the function values f of a point are the first NX entries of its compact record (lowering.py writes the variable entries in the order
// of the dense layout, and f comes first): lane b reads f_b at a lane-dependent address
constexpr bool qd_f_linear() {
  for (int i = 0; i < NX; ++i) if (DOMPC_DYN_VIDX[i] != i) return false;
  return true;
}

// the indices of this lane's edge (KArgs::edge_pack); `om` = omega * objective scaling
struct QdPack { int woff, row0, xoffp, xoffc, level, epsoff; double om; };
__device__ inline QdPack qd_pack(const KArgs& A, int e, double sf) {
  const auto* ep = A.edge_pack + e * EP_N;
  QdPack k;
  k.woff = ep[EP_WOFF]; k.row0 = ep[EP_ROW0]; k.xoffp = ep[EP_XOFF_PARENT]; k.xoffc = ep[EP_XOFF_CHILD]; k.level = ep[EP_LEVEL]; k.epsoff = ep[EP_EPSOFF_PARENT];
  k.om = __builtin_bit_cast(double, ((unsigned long long)(unsigned)ep[EP_OMEGA_HI] << 32) | (unsigned long long)(unsigned)ep[EP_OMEGA_LO]) * sf;
  return k;
}
// request the compact records of edges e0 .. e0 + 3 (contiguous in memory) into bank `bank` of the wavefront's LDS region (LDS-DMA,
// 64 lanes x 16 B per instruction).  Runs past the last record of the problem into the arrays behind it (ws_layout) - never used.
__device__ inline void stage_quad(const Prob& Q, int e0, int lane, ldsd* Ld, int bank) {
  const double* src = Q.mo + (int64_t)e0 * MO_REC;
  ldsd* dst = Ld + bank * QL_MOSZ;
#pragma unroll
  for (int q = 0; q < QL_MOSZ / 128; ++q)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 128 * q + 2 * lane),
                                     (__attribute__((address_space(3))) void*)(dst + 128 * q), 16, 0, 0);
}

#ifndef DOMPC_QUAD_PROFILE
#define DOMPC_QUAD_PROFILE DOMPC_PROFILE
#endif
#ifndef DOMPC_QUAD_SB
#define DOMPC_QUAD_SB 1            // scheduling barriers between the pieces of a quad and between the pivots of its elimination: without them the
#endif                             // machine scheduler stretches live ranges over the whole 9 000-instruction body and spills hundreds of registers
#if DOMPC_QUAD_SB
#define QD_SB() __builtin_amdgcn_sched_barrier(0)
#else
#define QD_SB()
#endif

// Derivative evaluation, factorisation and condensing of the edges e0 .. e0 + 3 (edge e0 + (lane >> 4) on each row of 16 lanes; rows
// beyond the last edge repeat it and store nothing).  `bank`: where stage_quad() put their compact records; `e0n`: first edge of the
// quad this wavefront handles next (-1: none) - requested into the other bank once the elimination is through.
// Returns 0, 1 (a singular block cannot happen here: it takes the fallback) or 2: at least one collocation block failed the pivot test
// in its natural order - nothing that the fallback (eval_edge_coop on each edge) does not write again has been stored.
__device__ inline int eval_edge_quad(const Thr& T, const Prob& Q, int e0, int e0n, double mu, int lane_, ldsd* Ld, int bank, const QdPack& pk, QdPack& pkn, bool store_lu) {
  const KArgs& A = *Q.A;
  // (the lane number as a value the optimiser cannot see through: everything derived from it - 16 lane predicates, the per-lane selects of
  //  collocation coefficients and unit vectors - is otherwise hoisted out of the quad loop as loop-invariant and held in ~100 registers)
  int lane = lane_;
  asm volatile("" : "+v"(lane));
  constexpr int R = DEG * NX;
  constexpr double GJ_U = DOMPC_GJ_U;
  const int g = lane >> 4, j = lane & 15;
  const bool act = e0 + g < A.n_edges;
  const int e = act ? e0 + g : A.n_edges - 1;
  const bool isx = j < NX;
  const bool st_x = act && isx;                       // this lane stores per-state results
  const unsigned b = (unsigned)(isx ? j : 0);
  const bool soc = (Q.soc & 1) != 0;
  const double omh = (Q.soc & 2) ? 0.0 : pk.om;       // weight of the objective HESSIANS (Prob::soc bit 1)
#if DOMPC_QUAD_PROFILE
  long long pc0 = prof_clock();
#define QD_PH(i) if (T.prof && T.tid == 0) { const long long pc1 = prof_clock(); T.prof[i] += pc1 - pc0; pc0 = pc1; }
#else
#define QD_PH(i)
#endif
  // ---- 1. operands outside the model-output record: one batch of loads (iterate, bounds, bound multipliers, row multipliers)
  const double xn = ldoff(Q.x, (unsigned)pk.xoffp + b), xc = ldoff(Q.x, (unsigned)pk.xoffc + b);
  double wv[M], lbv[M], ubv[M], zlv[M], zuv[M];
#pragma unroll
  for (int s = 0; s < M; ++s) {
    const unsigned gi = (unsigned)pk.woff + (unsigned)(s * NX) + b;
    wv[s] = ldoff(Q.x, gi); lbv[s] = ldoff(Q.lb, gi); ubv[s] = ldoff(Q.ub, gi); zlv[s] = ldoff(Q.zl, gi); zuv[s] = ldoff(Q.zu, gi);
  }
  double lam[DEG], cin[DEG];
#pragma unroll
  for (int jj = 0; jj < DEG; ++jj) {
    lam[jj] = ldoff(Q.lam, (unsigned)pk.row0 + (unsigned)(jj * NX) + b);
    cin[jj] = soc ? ldoff(Q.c, (unsigned)pk.row0 + (unsigned)(jj * NX) + b) : 0.0;
  }
  const double lamc = ldoff(Q.lam, (unsigned)pk.row0 + (unsigned)R + b), nue = ldoff(Q.lam, (unsigned)pk.row0 + (unsigned)NW + b);
  const double cinc = soc ? ldoff(Q.c, (unsigned)pk.row0 + (unsigned)R + b) : 0.0, cine = soc ? ldoff(Q.c, (unsigned)pk.row0 + (unsigned)NW + b) : 0.0;
  // nl_cons rows of the edge (one evaluation of NE rows at (x_n, u_n), _mpc.py:1239-1246): multipliers, row scalings, slack variables
  // with their bounds and bound multipliers, the eps entries the rows read - the same for every lane of the edge
  double ydv[NE1], sgv[NE1], slv[NE1], sllv[NE1], sluv[NE1], zslv[NE1], zsuv[NE1], cnl[NE1], epv[NSE > 0 ? NSE : 1];
  if constexpr (NE > 0) {
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const unsigned si = (unsigned)(e * NE1 + i);
      ydv[i] = ldoff(Q.lam, (unsigned)pk.row0 + (unsigned)(NW + NX + i));
      sgv[i] = ldoff(Q.sgn, si); slv[i] = ldoff(Q.s, si); sllv[i] = ldoff(Q.sl, si); sluv[i] = ldoff(Q.su, si);
      zslv[i] = ldoff(Q.zsl, si); zsuv[i] = ldoff(Q.zsu, si);
      cnl[i] = soc ? ldoff(Q.c, (unsigned)pk.row0 + (unsigned)(NW + NX + i)) : 0.0;
    }
#pragma unroll
    for (int q = 0; q < NSE; ++q) epv[q] = ldoff(Q.x, (unsigned)pk.epsoff + (unsigned)q);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the staged records (and everything above) have landed
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  QD_PH(4)
  QD_SB();
  const ldsd* rec = Ld + bank * QL_MOSZ + g * MO_REC;         // compact record of this lane's edge
  // ---- 2. residual rows of state b: collocation rows (one per point), continuity row, end-point row (optimizer.py:951-983, _mpc.py:1224)
  double res[DEG], rc, ce;
  {
    double fv[DEG];
    if constexpr (qd_f_linear()) {
#pragma unroll
      for (int jj = 0; jj < DEG; ++jj) fv[jj] = rec[jj * DOMPC_DYN_NV + (int)b];
    } else {
      sfor<DEG>([&](auto JJ) {
        constexpr int jj = JJ;
        double v = 0.0;
        sfor<NX>([&](auto B_) {
          constexpr int bb = B_;
          constexpr MoRef rf = moref_dyn(jj, bb);
          if constexpr (rf.kind != 0) { const double t_ = mo_dyn_val<rf.kind, rf.off>(rec); v = (j == bb) ? t_ : v; }
        });
        fv[jj] = v;
      });
    }
#pragma unroll
    for (int jj = 0; jj < DEG; ++jj) {
      double xp = DOMPC_C[0 * (DEG + 1) + jj + 1] * xn;
#pragma unroll
      for (int r = 1; r <= DEG; ++r) xp += DOMPC_C[r * (DEG + 1) + jj + 1] * wv[r - 1];
      res[jj] = soc ? cin[jj] : fv[jj] - xp;
    }
    double xf = DOMPC_D[0] * xn;
#pragma unroll
    for (int r = 1; r <= DEG; ++r) xf += DOMPC_D[r] * wv[r - 1];
    rc = soc ? cinc : wv[M - 1] - xf;
    ce = soc ? cine : wv[M - 1] - xc;
  }
  // barrier terms of this lane's unknowns (one per slot): gradient at mu and per unit mu, Sigma, -z_L + z_U.  Before the columns are built:
  // the iterate, the bounds and their multipliers (15 values) are dead afterwards
  double BGv[M], BBv[M], SGv[M], ZDv[M];
#pragma unroll
  for (int s = 0; s < M; ++s) {
    const double xv = wv[s], l = lbv[s], u = ubv[s];
    BGv[s] = bar_grad(xv, l, u, mu, !(Q.soc & 2));
    BBv[s] = bar_grad(xv, l, u, 1.0);
    SGv[s] = sigma_of(xv, l, u, zlv[s], zuv[s]);
    ZDv[s] = zuv[s] - zlv[s];
  }
#pragma unroll
  for (int s = 0; s < M; ++s) { pin(BGv[s]); pin(BBv[s]); pin(SGv[s]); pin(ZDv[s]); }
#pragma unroll
  for (int jj = 0; jj < DEG; ++jj) pin(res[jj]);
  pin(rc); pin(ce);
  // ---- 3. this lane's columns of [G_cc | J_u | r]: bc[s][(jj, a)]
  //   lane b < NX, slot s: column (s, b) of G_cc = [s == jj] J_jj[a][b] - [a == b] C[s+1][jj+1]
  //   lanes NX .. NA-1, slot 0: column ku of the input Jacobians J_jj[a][NX + ku];  lane NA, slot 0: the residual rows;  zero otherwise
  double bc[DEG][R];
  sfor<DEG>([&](auto S_) {
    constexpr int s = S_;
    sfor<R>([&](auto R_) {
      constexpr int r = R_, jj = r / NX, a = r % NX;
      double v = 0.0;
      if constexpr (s == jj)
        sfor<NX>([&](auto B_) {
          constexpr int bb = B_;
          constexpr MoRef rf = moref_dyn(jj, NX + a * NA + bb);
          if constexpr (rf.kind != 0) { const double t_ = mo_dyn_val<rf.kind, rf.off>(rec); v = (j == bb) ? t_ : v; }
        });
      if constexpr (s == 0) {
        sfor<NU>([&](auto K_) {
          constexpr int ku = K_;
          constexpr MoRef rf = moref_dyn(jj, NX + a * NA + NX + ku);
          if constexpr (rf.kind != 0) { const double t_ = mo_dyn_val<rf.kind, rf.off>(rec); v = (j == NX + ku) ? t_ : v; }
        });
        const double rb_ = rbc<a>(res[jj]);       // (the broadcast is executed by every lane: never inside a select's branch)
        v = (j == NA) ? rb_ : v;
      }
      v = (j == a) ? v - DOMPC_C[(s + 1) * (DEG + 1) + jj + 1] : v;
      bc[s][r] = v;
    });
  });
  QD_PH(5)
  QD_SB();
  // ---- 4. dual-residual pieces: column times the multipliers of the edge's rows
  double RWv[M], RDv[M], ry;
  {
    double tcol[M];
#pragma unroll
    for (int s = 0; s < M; ++s) tcol[s] = 0.0;
    double ryu = 0.0;
    // multipliers of the collocation rows (jj, a) from their owner, lane a: block (jj, jj) of G_cc and the input columns
    sfor<DEG>([&](auto J_) {
      constexpr int jj = J_;
      sfor<NX>([&](auto A_) {
        constexpr int a = A_;
        const double la = rbc<a>(lam[jj]);
        tcol[jj] = fma(la, bc[jj][jj * NX + a], tcol[jj]);       // J_jj' lambda_jj - C[jj+1][jj+1] lambda_jj
        if constexpr (jj == 0) ryu = fma(la, bc[0][a], ryu);
        else ryu = fma(la, bc[0][jj * NX + a], ryu);
      });
    });
#pragma unroll
    for (int s = 0; s < DEG; ++s) {
      double t = tcol[s];
#pragma unroll
      for (int jj = 0; jj < DEG; ++jj)
        if (jj != s) t = fma(-DOMPC_C[(s + 1) * (DEG + 1) + jj + 1], lam[jj], t);
      tcol[s] = t - DOMPC_D[s + 1] * lamc;
    }
    tcol[M - 1] = lamc + nue;                        // end-point column: +1 in its continuity row, +1 in the end-point row
    // parent state columns: -C[0][j] in row (jj, b), -D_0 in the continuity row; input columns: J_u' lambda
    double ryx = -DOMPC_D[0] * lamc;
#pragma unroll
    for (int jj = 0; jj < DEG; ++jj) ryx = fma(-DOMPC_C[0 * (DEG + 1) + jj + 1], lam[jj], ryx);
    ry = isx ? ryx : ryu;                             // (lanes < NA; completed with the cost gradient below)
#pragma unroll
    for (int s = 0; s < M; ++s) { RWv[s] = tcol[s] + BGv[s]; RDv[s] = tcol[s] + ZDv[s]; }
#pragma unroll
    for (int s = 0; s < M; ++s) { pin(RWv[s]); pin(RDv[s]); }
    pin(ry);
    if (isx) {
      ldsd* Lv = Ld + QL_RW + g * QL_VG;
#pragma unroll
      for (int s = 0; s < M; ++s) { Lv[s * NX + (int)b] = RWv[s]; Lv[NW + s * NX + (int)b] = BBv[s]; }
    }
  }
  QD_PH(6)
  QD_SB();
  // ---- 5. in-place Gauss-Jordan in the natural pivot order: column k = (sk, bk) lives in slot sk of lane bk
  // The pivot column is replaced by column k of the inverse in place; its scaling -1 / a_kk is DEFERRED: the lane keeps the column as it
  // is (row k: -1) and remembers the factor of that slot - every later update is linear in the column, so one multiplication per entry at
  // the end replaces one per entry and pivot (19 of the ~90 instructions of a step).
  // Pivot test: |a_kk| >= GJ_U max |a_rk| over the rows of the same group of four - the test of the blocked elimination this replaces
  // (edge_factor_mfma tests the multipliers of its 4 x 4 pivot blocks); a failure sends the quad through the fallback.
  int bad = 0;
  double sig[DEG];
#pragma unroll
  for (int s = 0; s < DEG; ++s) sig[s] = 1.0;
  sfor<R>([&](auto K_) {
    constexpr int k = K_, sk = k / NX, bk = k % NX;
    constexpr int rt1 = (4 * (k / 4 + 1) < R) ? 4 * (k / 4 + 1) : R;
    QD_SB();
    const bool own = (j == bk);
    double m = 0.0;
#pragma unroll
    for (int r = k + 1; r < rt1; ++r) m = fmax(m, fabs(bc[sk][r]));
    const double akk = fabs(bc[sk][k]);
    bad |= (int)(own & !(akk >= GJ_U * m && akk > 1e-300));
    const double pl = fast_rcp((akk > 1e-300) ? bc[sk][k] : 1.0);
    const double pinv = rbc<bk>(pl);
    sig[sk] = own ? -pl : sig[sk];
    double pm[DEG];                                   // minus the scaled pivot row
#pragma unroll
    for (int s = 0; s < DEG; ++s) {
      const double prow = bc[s][k] * pinv;
      pm[s] = (s == sk && own) ? 0.0 : -prow;         // (the pivot column itself keeps its entries)
      bc[s][k] = (s == sk && own) ? -1.0 : prow;      // row k: scaled; the pivot position: 1 / a_kk = (-1 / a_kk) (-1)
      pin(pm[s]);
    }
    // rows r != k: a_r. -= a_rk (a_k. / a_kk), a_rk read from its owner through the DPP operand.  The slot of the pivot column LAST: its
    // update overwrites the broadcast source (and nothing of this step wrote that register before - the hazard rule of fmac_rbc)
    asm volatile("s_nop 1");
    sfor<R>([&](auto R_) {
      constexpr int r = R_;
      if constexpr (r != k) {
        sfor<DEG>([&](auto S_) { constexpr int s = S_; if constexpr (s != sk) fmac_rbc<bk>(bc[s][r], bc[sk][r], pm[s]); });
        fmac_rbc<bk>(bc[sk][r], bc[sk][r], pm[sk]);
      }
    });
    // (every entry is computed HERE: without this the updates of the last pivots are sunk behind the elimination, next to their first
    //  use, and drag the broadcast columns of five steps along - 250 live registers at the end of the elimination)
#pragma unroll
    for (int s = 0; s < DEG; ++s)
#pragma unroll
      for (int r = 0; r < R; ++r) pin(bc[s][r]);
  });
  QD_SB();
#pragma unroll
  for (int s = 0; s < DEG; ++s)
#pragma unroll
    for (int r = 0; r < R; ++r) bc[s][r] *= sig[s];
  QD_SB();
  if (__ballot(bad) != 0ull) return 2;
#ifdef QD_CUT
  if (st_x) { double t = 0; for (int s = 0; s < DEG; ++s) for (int r = 0; r < R; ++r) t += bc[s][r]; Q.ew[(int64_t)e * EW_SIZE + (int)b] = t + RWv[0] + SGv[1] + ry + rc + ce; }
  return 0;
#endif
  QD_PH(2)
  QD_SB();
  // now: lane b < NX, slot s: column (s, b) of G_cc^-1;  lanes NX .. NA-1, slot 0: G_cc^-1 J_u;  lane NA, slot 0: G_cc^-1 r
  // ---- 6. forward-pass record: G_cc^-1 (row-major), Sigma_w, r_w
  if (st_x) {
    if (!soc) {
#pragma unroll
      for (int jj = 0; jj < DEG; ++jj) Q.c[pk.row0 + jj * NX + (int)b] = res[jj];
      Q.c[pk.row0 + R + (int)b] = rc;
      Q.c[pk.row0 + NW + (int)b] = ce;
    }
#pragma unroll
    for (int s = 0; s < M; ++s) {
      const int gi = pk.woff + s * NX + (int)b;
      Q.gf[gi] = 0.0;
      Q.rd[gi] = RDv[s];
    }
    double* ew = Q.ew + (int64_t)e * EW_SIZE;
    if (store_lu) {                                  // (uniform, lu_store_rule; QUAD_FWD: the forward pass forms the inverse again - 400 of the record's 464 doubles stay home)
#pragma unroll
      for (int s = 0; s < DEG; ++s)
#pragma unroll
        for (int r = 0; r < R; ++r) ew[EW_LU + r * LU_N + s * NX + (int)b] = bc[s][r];
    }
#pragma unroll
    for (int s = 0; s < M; ++s) {
      ew[EW_SIGW + s * NX + (int)b] = SGv[s] + Q.dsw;
      ew[EW_RW + s * NX + (int)b] = RWv[s];
    }
  }
  // the compact records of the quad this wavefront handles next: on their way while the condensing runs
  if (e0n >= 0) stage_quad(Q, e0n, lane, Ld, bank ^ 1);
  {
    const int en = (e0n >= 0 ? e0n : e0) + g;                   // (its indices: requested now, needed at the top of the next quad)
    pkn = qd_pack(A, en < A.n_edges ? en : A.n_edges - 1, Q.sf);
  }
  // ---- 7. this lane's column of [W | w0]: W_x = sum_s C[0][s+1] (G_cc^-1)[:, (s, b)], W_u = -G_cc^-1 J_u, w0 = -G_cc^-1 r; continuity rows
  double Wc[NW];
  {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double t = (isx ? DOMPC_C[0 * (DEG + 1) + 1] : -1.0) * bc[0][r];
#pragma unroll
      for (int s = 1; s < DEG; ++s) t = fma(isx ? DOMPC_C[0 * (DEG + 1) + s + 1] : 0.0, bc[s][r], t);
      Wc[r] = t;
    }
    sfor<NX>([&](auto A_) {
      constexpr int a = A_;
      const double rcb = rbc<a>(rc);
      double t = (j == NA) ? -rcb : ((j == a) ? DOMPC_D[0] : 0.0);
#pragma unroll
      for (int s = 1; s <= DEG; ++s) t = fma(DOMPC_D[s], Wc[(s - 1) * NX + a], t);
      Wc[R + a] = t;
    });
  }
#pragma unroll
  for (int r = 0; r < NW; ++r) pin(Wc[r]);
  double* S_ = Q.es + (int64_t)e * ES_SIZE;
  {
    // linearised dynamics of the interval: [A B] = end-point rows of W, c~ = w0_end + (end-point residual)
    double cv = 0.0;
    sfor<NX>([&](auto A_) {
      constexpr int a = A_;
      if (act && j < NA) S_[ES_AB + a * NA + j] = Wc[R + a];
      const double w0a = rbc<NA>(Wc[R + a]);
      cv = (j == a) ? w0a + ce : cv;
    });
    pin(cv);
    if (st_x) S_[ES_CV + (int)b] = cv;
  }
  QD_PH(7)
  QD_SB();
  // ---- 8. condensing.  Lane c owns column c of Z~ = [W | w0 | 0]:  acc[i] = row i of
  //      [Q~ | q~ | W'b] = sum_p Z_p' ((H_p + Sigma_p) Z~_p + [0 | r_w,p | b_p]) + W_k' (Sigma_k W~_k + [0 | r_w,k | b_k])
  double acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = 0.0;
  {
    const double ind = (j >= NA) ? 1.0 : 0.0;                 // the two vector columns (lanes NA, NA + 1)
    ldsd* Wb = Ld + QL_WB + g * QL_WBG;
    const ldsd* Lvec = Ld + QL_RW + g * QL_VG + (j == NA + 1 ? NW : 0);
    sfor<M>([&](auto P_) {
      constexpr int p = P_;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (j <= NA) {
#pragma unroll
        for (int k = 0; k < NX; ++k) Wb[k * QL_WS + j] = Wc[p * NX + k];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      double Zc[NX], Tv[NA];
#pragma unroll
      for (int k = 0; k < NX; ++k) Zc[k] = (j <= NA) ? Wc[p * NX + k] : 0.0;
#pragma unroll
      for (int i = 0; i < NA; ++i) Tv[i] = 0.0;
      if constexpr (p < DEG) {
        // lambda-weighted Hessian of the dynamics at point p over (x_p, u): packed upper triangle, only its structural non-zeros
        sfor<NA>([&](auto I_) {
          constexpr int i = I_;
          sfor<NA - i>([&](auto D_) {
            constexpr int k = i + D_;
            constexpr MoRef rf = moref_dyn(p, MOH_H0 + symi(i, k, NA));
            if constexpr (rf.kind != 0) {
              const double h = mo_dyn_val<rf.kind, rf.off>(rec);
              // entry (i, k) and its mirror; rows / columns >= NX belong to u: Z~ has the unit vector e_c there (columns NX .. NA-1)
              if constexpr (k < NX) {
                Tv[i] = fma(h, Zc[k], Tv[i]);
                if constexpr (i != k) Tv[k] = fma(h, Zc[i], Tv[k]);
              } else {
                Tv[i] += (j == k) ? h : 0.0;
                if constexpr (i != k) {
                  if constexpr (i < NX) Tv[k] = fma(h, Zc[i], Tv[k]);
                  else Tv[k] += (j == i) ? h : 0.0;
                }
              }
            }
          });
        });
      }
      sfor<NX>([&](auto K_) {
        constexpr int k = K_;
        const double sg = rbc<k>(SGv[p]) + Q.dsw;
        Tv[k] = fma(sg, Zc[k], Tv[k]);
        Tv[k] = fma(ind, Lvec[p * NX + k], Tv[k]);
      });
      // acc[i] += sum_k Z_p[k][i] T[k]: the rows of Z_p from their staged copy, two rows per batch of LDS reads (the scheduler would
      // otherwise issue all 130 reads of the slot up front and spill what they displace)
      sfor<(NX + 1) / 2>([&](auto K2_) {
        constexpr int k0 = 2 * K2_;
        QD_SB();
        double wr[2][NA];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < NA; ++i) wr[kk][i] = (k0 + kk < NX) ? (double)Wb[(k0 + kk) * QL_WS + i] : 0.0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          if (k0 + kk < NX) {
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i] = fma(wr[kk][i], Tv[k0 + kk], acc[i]);
          }
      });
      if constexpr (p < DEG) {
#pragma unroll
        for (int ku = 0; ku < NU; ++ku) acc[NX + ku] += Tv[NX + ku];
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) asm volatile("" : "+v"(acc[i]));
      QD_SB();
    });
  }
  // stage-cost Hessian (packed in the record; only its structural non-zeros)
  sfor<NA>([&](auto I_) {
    constexpr int i = I_;
    sfor<NA - i>([&](auto D_) {
      constexpr int k = i + D_;
      constexpr MoRef rf = moref_lt(1 + NA + symi(i, k, NA));
      if constexpr (rf.kind != 0) {
        const double h = omh * mo_lt_val<rf.kind, rf.off>(rec);
        acc[i] += (j == k) ? h : 0.0;
        if constexpr (i != k) acc[k] += (j == i) ? h : 0.0;
      }
    });
  });
  if constexpr (NE > 0) {
    // lambda-weighted Hessian of the nl_cons rows over (x_n, u_n) (packed; only its structural non-zeros)
    sfor<NA>([&](auto I_) {
      constexpr int i = I_;
      sfor<NA - i>([&](auto D_) {
        constexpr int k = i + D_;
        constexpr MoRef rf = moref_nl(NE + NE * NA + symi(i, k, NA));
        if constexpr (rf.kind != 0) {
          const double h = mo_nl_val<rf.kind, rf.off>(rec);
          acc[i] += (j == k) ? h : 0.0;
          if constexpr (i != k) acc[k] += (j == i) ? h : 0.0;
        }
      });
    });
  }
  QD_PH(1)
  QD_SB();
  // ---- 9. shared record of the edge
  {
    ldsd* Lq = Ld + QL_QV + g * 32;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (j == NA || j == NA + 1) {
#pragma unroll
      for (int i = 0; i < NA; ++i) Lq[(j - NA) * 16 + i] = acc[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    sfor<NA>([&](auto I_) {
      constexpr int i = I_;
      if (act && j >= i && j < NA) S_[ES_QT + symi(i, i, NA) + (j - i)] = acc[i];      // row i of the packed upper triangle: entries (i, i .. NA-1)
    });
    // gradient of the stage cost w.r.t. (x_n, u_n): entry j
    double ltg = 0.0;
    sfor<NA>([&](auto A_) {
      constexpr int a = A_;
      constexpr MoRef rf = moref_lt(1 + a);
      if constexpr (rf.kind != 0) { const double t_ = mo_lt_val<rf.kind, rf.off>(rec); ltg = (j == a) ? t_ : ltg; }
    });
    double rnl = 0.0;                                 // Jd' (y_d sg): share of the nl_cons rows in the dual residual of (x_n, u_n), entry j
    if constexpr (NE > 0) {
      sfor<NE>([&](auto I_) {
        constexpr int i = I_;
        double jd = 0.0;
        sfor<NA>([&](auto A_) {
          constexpr int a = A_;
          constexpr MoRef rf = moref_nl(NE + i * NA + a);
          if constexpr (rf.kind != 0) { const double t_ = mo_nl_val<rf.kind, rf.off>(rec); jd = (j == a) ? t_ : jd; }
        });
        const double jds = jd * sgv[i];
        rnl = fma(jds, ydv[i], rnl);
        if (act && j < NA) Q.ew[(int64_t)e * EW_SIZE + EW_JD + i * NA + j] = jds;
      });
      if (act && j == 0) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
          double d = 0.0;
          sfor<NE>([&](auto I_) {                      // (row i of d(x_n, u_n): its entry of the record)
            constexpr int i2 = I_;
            constexpr MoRef rf = moref_nl(i2);
            if constexpr (rf.kind != 0) { const double t_ = mo_nl_val<rf.kind, rf.off>(rec); d = (i == i2) ? t_ : d; }
          });
          const int sq = nl_slack(i);
          if (NSE > 0 && sq >= 0) {
#pragma unroll
            for (int q = 0; q < NSE; ++q) d -= (q == sq) ? epv[q] : 0.0;
          }
          d *= sgv[i];
          const double rdn = soc ? cnl[i] : d - slv[i];
          if (!soc) Q.c[pk.row0 + NW + NX + i] = rdn;
          S_[ES_RDN + i] = rdn;
          S_[ES_SIGS + i] = sigma_of(slv[i], sllv[i], sluv[i], zslv[i], zsuv[i]);
          S_[ES_RSN + i] = -ydv[i] + bar_grad(slv[i], sllv[i], sluv[i], mu);
        }
      }
    }
    if (act && j < NA) {
      const double gy = pk.om * ltg, r = ry + gy + rnl;
      S_[ES_GFY + j] = gy;
      S_[ES_RY + j] = r;
      S_[ES_QV + j] = Lq[j] + r;
      S_[ES_QVB + j] = Lq[16 + j];
    }
    const bool last = pk.level == A.N - 1;
    if (__ballot(last) != 0ull) {
      double mg = 0.0;
      sfor<NX>([&](auto A_) {
        constexpr int a = A_;
        constexpr MoRef rf = moref_mt(1 + a);
        if constexpr (rf.kind != 0) { const double t_ = mo_mt_val<rf.kind, rf.off>(rec); mg = (j == a) ? t_ : mg; }
      });
      if (st_x && last) S_[ES_MG + (int)b] = pk.om * mg;
      sfor<NX>([&](auto A_) {
        constexpr int a = A_;
        double v = 0.0;
        sfor<NX>([&](auto B_) {
          constexpr int bb = B_;
          constexpr MoRef rf = moref_mt(1 + NX + symi(a, bb, NX));
          if constexpr (rf.kind != 0) { const double t_ = mo_mt_val<rf.kind, rf.off>(rec); v = (j == bb) ? t_ : v; }
        });
        if (st_x && last) S_[ES_MH + a * NX + (int)b] = omh * v;
      });
    }
    if (act && j == 0) {
      constexpr MoRef r0 = moref_lt(0), m0 = moref_mt(0);
      double obj = pk.om * mo_lt_val<r0.kind, r0.off>(rec);
      if (last) obj += pk.om * mo_mt_val<m0.kind, m0.off>(rec);
      if constexpr (NE > 0) {
#pragma unroll
        for (int q = 0; q < NSE; ++q) obj += Q.sf * DOMPC_EPS_PEN[q] * epv[q];
      }
      S_[ES_OBJ] = obj;
    }
  }
  QD_PH(3)
  QD_SB();
#undef QD_PH
  return 0;
}

// the fallback of a quad whose pivot test failed: its edges one after the other through the wavefront-per-edge path (its own function:
// the code of that path - dense image, matrix-core elimination with its pivoting repeat - stays out of the quad loop's register allocation)
__device__ __attribute__((noinline)) int phase_edge_fallback(const void* kp, int slot, int b_, int e0, int soc, double sf, double mu, double dsw) {
  const KArgs A = kernel_args(kp);
  Thr T = make_thr(A);
  T.kp = kp;
  Prob Q = make_prob(A, __builtin_amdgcn_readfirstlane(slot), A.p + (int64_t)__builtin_amdgcn_readfirstlane(b_) * A.n_opt_p);
  Q.sf = ufl(sf);
  Q.soc = __builtin_amdgcn_readfirstlane(soc);
  Q.dsw = ufl(dsw);
  prob_bounds(Q);
  const int lane = (int)(threadIdx.x & 63u);
  ldsd* Ld = (ldsd*)lds_pool + (int64_t)(threadIdx.x >> 6) * EL_SIZE;
  const int ef = __builtin_amdgcn_readfirstlane(e0);
  const MocMap mm = moc_map(lane, 64);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  mo_image_init(Ld + EL_MOS, lane, 64);
  if (MFMA_GJ) gj_table_init(Ld, lane);
  int staged = -1, fail = 0;
  for (int q = 0; q < 4; ++q) {
    const int e = ef + q;
    if (e < A.n_edges) fail |= eval_edge_coop(T, Q, e, -1, ufl(mu), lane, 64, Ld, staged, mm);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return fail;
}

// the edge loop of the derivative sweep: quads of consecutive edges, dealt to the wavefronts of the problem round-robin.  Its own
// function (one call per sweep): inlined into phase_sweep the loop shared a register allocation with the model evaluation and the node
// assembly around it and spilled ~370 values per quad.
__device__ __attribute__((noinline)) int phase_sweep_quads(const void* kp, int b_, int slot, int soc, double sf, double mu, double dsw) {
  const KArgs A = kernel_args(kp);
  Thr T = make_thr(A);
  T.kp = kp;
  Prob Q = make_prob(A, ufl(slot), A.p + (int64_t)ufl(b_) * A.n_opt_p);
  Q.sf = ufl(sf);
  Q.soc = ufl(soc);
  Q.dsw = ufl(dsw);
  prob_bounds(Q);
  mu = ufl(mu);
  const int ng = T.nt / 64, gid = group_index(T.tid, 64), lane = T.tid % 64;
  ldsd* Ld = T.edge_lds + (int64_t)(T.ltid / 64) * EL_SIZE;
  const int nq = (A.n_edges + 3) / 4;
  int fail = 0, bank = 0;
  int qd = gid;
  if (qd >= nq) return 0;
  const bool store_lu = lu_store_rule(A, mu, Q.soc);
  const int gl = lane >> 4;
  auto pack_of = [&](int q) { const int e = 4 * q + gl; return qd_pack(A, e < A.n_edges ? e : A.n_edges - 1, Q.sf); };
  stage_quad(Q, 4 * qd, lane, Ld, 0);
  QdPack pk = pack_of(qd);
  while (qd < nq) {
    const int qn = qd + ng;
    const bool more = qn < nq;
    QdPack pkn = pk;
    const int rc = eval_edge_quad(T, Q, 4 * qd, more ? 4 * qn : -1, mu, lane, Ld, bank, pk, pkn, store_lu);
    if (__builtin_amdgcn_readfirstlane(rc) == 2) {
      pkn = pack_of(more ? qn : qd);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (no LDS-DMA in flight into the region the fallback uses)
      fail |= phase_edge_fallback(kp, Q.slot, b_, 4 * qd, Q.soc, Q.sf, mu, Q.dsw);
      if (more) stage_quad(Q, 4 * qn, lane, Ld, bank ^ 1);
    } else {
      fail |= rc;
    }
    bank ^= 1;
    pk = pkn;
    qd = qn;
  }
  return fail;
}
__device__ inline int sweep_quads(const Thr& T, const Prob& Q, double mu) {
  const KArgs& A = *Q.A;
  const int b_ = (int)((Q.P - A.p) / A.n_opt_p);
  return phase_sweep_quads(T.kp, b_, Q.slot, Q.soc, Q.sf, mu, Q.dsw);
}
#elif !defined(DOMPC_HOST_EMU)
__device__ inline int sweep_quads(const Thr&, const Prob&, double) { return 0; }      // (never called: QUAD_EDGE is false for such a model)
#endif

// ================================================================================================
// Forward pass, per-edge part, with four edges per wavefront (round 6): the steps of the collocation unknowns dw and of the multipliers of
// the edge's rows, dlambda, from the steps of the node variables (riccati_forward_t: chain walk / branching levels).  The wavefront-per-
// two-edges version (dompc_forward.h, FE2) reads the stored inverse G_cc^-1 of every edge (400 of the 464 doubles of the forward record:
// written by every sweep, read back by every forward pass - a sixth of the kernel's memory traffic).  Here the inverse is FORMED AGAIN from
// the compact model-output record, which the pass reads anyway (point Hessians, input Jacobians): the same columns and the same
// elimination as in the sweep (qd_fw_columns / qd_fw_eliminate restate steps 3 and 5 of eval_edge_quad; kept apart from it - shared
// helper functions changed the register allocation of the sweep's quad loop and brought its spills back), with the right-hand side
// g = G_y dy + r in the residual column, so that dw = -G_cc^-1 g comes out of the elimination itself.  The batch path is bound by its
// memory traffic, not by instruction issue (profiles/r06_backward4.txt): 1 500 more vector instructions per four edges cost nothing
// that 3.6 KB less traffic per edge does not repay.  The sweep stores the inverse only when the next forward pass may be the adjoint
// variant (last barrier levels), which still reads it, or when its pivot test failed (the fallback below).
#if !defined(DOMPC_HOST_EMU) && DOMPC_DEG >= 1 && DOMPC_M >= 1 && DOMPC_NI == 1 && DOMPC_NX + DOMPC_NU + 2 <= 16 && DOMPC_DEG * DOMPC_DEG * DOMPC_NX <= 64
static_assert(!QUAD_FWD || (2 * NW + NA + DEG * NX <= QF_VG && QL_WB + 4 * QF_VG <= EL_SIZE), "LDS of the four-edge forward pass");

__device__ inline void qd_fw_columns(const ldsd* rec, int j, const double (&res)[DEG], double (&bc)[DEG][DEG * NX]) {
  constexpr int R = DEG * NX;
  sfor<DEG>([&](auto S_) {
    constexpr int s = S_;
    sfor<R>([&](auto R_) {
      constexpr int r = R_, jj = r / NX, a = r % NX;
      double v = 0.0;
      if constexpr (s == jj)
        sfor<NX>([&](auto B_) {
          constexpr int bb = B_;
          constexpr MoRef rf = moref_dyn(jj, NX + a * NA + bb);
          if constexpr (rf.kind != 0) { const double t_ = mo_dyn_val<rf.kind, rf.off>(rec); v = (j == bb) ? t_ : v; }
        });
      if constexpr (s == 0) {
        sfor<NU>([&](auto K_) {
          constexpr int ku = K_;
          constexpr MoRef rf = moref_dyn(jj, NX + a * NA + NX + ku);
          if constexpr (rf.kind != 0) { const double t_ = mo_dyn_val<rf.kind, rf.off>(rec); v = (j == NX + ku) ? t_ : v; }
        });
        const double rb_ = rbc<a>(res[jj]);
        v = (j == NA) ? rb_ : v;
      }
      v = (j == a) ? v - DOMPC_C[(s + 1) * (DEG + 1) + jj + 1] : v;
      bc[s][r] = v;
    });
  });
}
__device__ inline int qd_fw_eliminate(int j, double (&bc)[DEG][DEG * NX]) {
  constexpr int R = DEG * NX;
  constexpr double GJ_U = DOMPC_GJ_U;
  int bad = 0;
  double sig[DEG];
#pragma unroll
  for (int s = 0; s < DEG; ++s) sig[s] = 1.0;
  sfor<R>([&](auto K_) {
    constexpr int k = K_, sk = k / NX, bk = k % NX;
    constexpr int rt1 = (4 * (k / 4 + 1) < R) ? 4 * (k / 4 + 1) : R;
    QD_SB();
    const bool own = (j == bk);
    double m = 0.0;
#pragma unroll
    for (int r = k + 1; r < rt1; ++r) m = fmax(m, fabs(bc[sk][r]));
    const double akk = fabs(bc[sk][k]);
    bad |= (int)(own & !(akk >= GJ_U * m && akk > 1e-300));
    const double pl = fast_rcp((akk > 1e-300) ? bc[sk][k] : 1.0);
    const double pinv = rbc<bk>(pl);
    sig[sk] = own ? -pl : sig[sk];
    double pm[DEG];
#pragma unroll
    for (int s = 0; s < DEG; ++s) {
      const double prow = bc[s][k] * pinv;
      pm[s] = (s == sk && own) ? 0.0 : -prow;
      bc[s][k] = (s == sk && own) ? -1.0 : prow;
      pin(pm[s]);
    }
    asm volatile("s_nop 1");
    sfor<R>([&](auto R_) {
      constexpr int r = R_;
      if constexpr (r != k) {
        sfor<DEG>([&](auto S_) { constexpr int s = S_; if constexpr (s != sk) fmac_rbc<bk>(bc[s][r], bc[sk][r], pm[s]); });
        fmac_rbc<bk>(bc[sk][r], bc[sk][r], pm[sk]);
      }
    });
#pragma unroll
    for (int s = 0; s < DEG; ++s)
#pragma unroll
      for (int r = 0; r < R; ++r) pin(bc[s][r]);
  });
  QD_SB();
#pragma unroll
  for (int s = 0; s < DEG; ++s)
#pragma unroll
    for (int r = 0; r < R; ++r) bc[s][r] *= sig[s];
  QD_SB();
  return bad;
}

struct QfPack { int woff, row0, level, n, cn, uoffp, epsoff; };
__device__ inline QfPack qf_pack(const KArgs& A, int e) {
  const auto* ep = A.edge_pack + e * EP_N;
  QfPack k;
  k.woff = ep[EP_WOFF]; k.row0 = ep[EP_ROW0]; k.level = ep[EP_LEVEL]; k.n = ep[EP_PARENT]; k.cn = ep[EP_CHILD];
  k.uoffp = ep[EP_UOFF_PARENT]; k.epsoff = ep[EP_EPSOFF_PARENT];
  return k;
}

__device__ __attribute__((noinline)) void phase_forward_quads(const void* kp, int b_, int slot, int soc, double sf, double delta_, int cl_) {
  const KArgs A = kernel_args(kp);
  Thr T = make_thr(A);
  T.kp = kp;
  Prob Q = make_prob(A, ufl(slot), A.p + (int64_t)ufl(b_) * A.n_opt_p);
  Q.sf = ufl(sf);
  Q.soc = ufl(soc);
  prob_bounds(Q);
  const double delta = ufl(delta_);
  const int cl = ufl(cl_);
  (void)delta;
  constexpr int R = DEG * NX;
  const int ng = T.nt / 64, gid = group_index(T.tid, 64);
  ldsd* Ld = T.edge_lds + (int64_t)(T.ltid / 64) * EL_SIZE;
  const int nq = (A.n_edges + 3) / 4;
  int bank = 0;
  int qd = gid;
  if (qd >= nq) return;
  {
    const int lane0 = (int)(threadIdx.x & 63u);
    stage_quad(Q, 4 * qd, lane0, Ld, 0);
  }
  while (qd < nq) {
    int lane = (int)(threadIdx.x & 63u);
    asm volatile("" : "+v"(lane));
    const int g = lane >> 4, j = lane & 15;
    const int e0 = 4 * qd;
    const bool act = e0 + g < A.n_edges;
    const int e = act ? e0 + g : A.n_edges - 1;
    const bool isx = j < NX, st_x = act && isx;
    const int b = isx ? j : 0;
    const QfPack pk = qf_pack(A, e);
    const bool chain = pk.level >= cl;
    // ---- operands: steps of the parent node's variables, the residuals of the edge's rows, r_w and Sigma_w of its unknowns, d nu of its
    //      end-point rows (chain levels: from the chain walk; branching levels: P dx + p of the child node, stored here)
    const double dxn = Q.nd[(int64_t)pk.n * ND_SIZE + ND_DXT + b];
    double du[NU > 0 ? NU : 1];
#pragma unroll
    for (int u = 0; u < NU; ++u) du[u] = Q.dx[pk.uoffp + u];
    double cres[DEG], rwv[M], sgv[M];
#pragma unroll
    for (int jj = 0; jj < DEG; ++jj) cres[jj] = Q.c[pk.row0 + jj * NX + b];
    const double ccont = Q.c[pk.row0 + R + b];
    const double* ew = Q.ew + (int64_t)e * EW_SIZE;
#pragma unroll
    for (int s = 0; s < M; ++s) { rwv[s] = ew[EW_RW + s * NX + b]; sgv[s] = ew[EW_SIGW + s * NX + b]; }
    double dnu = Q.dlam[pk.row0 + NW + b];
    if (__ballot(!chain) != 0ull) {
      const double* Nc = Q.nd + (int64_t)pk.cn * ND_SIZE;
      double t = Nc[ND_PV + b];
#pragma unroll
      for (int bb = 0; bb < NA; ++bb) t = fma(Nc[ND_P + b * NA + bb], Nc[ND_DXT + bb], t);
      dnu = chain ? dnu : t;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the staged records (and everything above) have landed
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const ldsd* rec = Ld + bank * QL_MOSZ + g * MO_REC;
    ldsd* Lv = Ld + QL_WB + g * QF_VG;                        // dw | dy | rhs | rr of this lane's edge
    constexpr int V_DW = 0, V_DY = NW, V_RHS = NW + NA, V_RR = 2 * NW + NA;
    // dy -> LDS (the u part by the lanes NX ..: they hold du[j - NX] like everybody)
    if (isx) Lv[V_DY + j] = dxn;
    sfor<NU>([&](auto U_) { constexpr int u = U_; if (j == NX + u) Lv[V_DY + NX + u] = du[u]; });
    // ---- g = G_y dy + r on the collocation rows (jj, b): r - C[0][j] dx_n + J_u du (row b of the input Jacobian of point jj)
    double gv[DEG];
    sfor<DEG>([&](auto J_) {
      constexpr int jj = J_;
      double t = cres[jj] - DOMPC_C[0 * (DEG + 1) + jj + 1] * dxn;
      sfor<NX>([&](auto A_) {
        constexpr int a = A_;
        sfor<NU>([&](auto U_) {
          constexpr int u = U_;
          constexpr MoRef rf = moref_dyn(jj, NX + a * NA + NX + u);
          if constexpr (rf.kind != 0) { const double t_ = mo_dyn_val<rf.kind, rf.off>(rec) * du[u]; t += (j == a) ? t_ : 0.0; }
        });
      });
      gv[jj] = t;
      pin(gv[jj]);
    });
    QD_SB();
    double bc[DEG][R];
    qd_fw_columns(rec, j, gv, bc);
    QD_SB();
    const int bad = qd_fw_eliminate(j, bc);
    const bool fb = __ballot(bad) != 0ull;                   // (the sweep took its fallback for this quad and stored the inverses: used below)
    // the compact records of the quad this wavefront handles next
    const int qn = qd + ng;
    if (qn < nq) stage_quad(Q, 4 * qn, lane, Ld, bank ^ 1);
    // ---- dw: collocation slots from lane NA's column -G_cc^-1 g, end-point slot from the continuity rows
    double dwv[M];
    if (!fb) {
#pragma unroll
      for (int s = 0; s < DEG; ++s) dwv[s] = 0.0;
      sfor<R>([&](auto R_) {
        constexpr int r = R_, s = r / NX, a = r % NX;
        const double t_ = rbc<NA>(bc[0][r]);
        dwv[s] = (j == a) ? -t_ : dwv[s];
      });
    } else {
      // fallback: dw = -G_cc^-1 g with the inverse the sweep's fallback stored (row (s, b) of it, g handed round through LDS)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (isx) {
#pragma unroll
        for (int s = 0; s < DEG; ++s) Lv[V_RR + s * NX + j] = gv[s];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < DEG; ++s) {
        double t = 0.0;
        for (int k2 = 0; k2 < R; ++k2) t -= ew[EW_LU + (s * NX + b) * LU_N + k2] * Lv[V_RR + k2];
        dwv[s] = t;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    {
      double t = DOMPC_D[0] * dxn - ccont;
#pragma unroll
      for (int s = 1; s <= DEG; ++s) t = fma(DOMPC_D[s], dwv[s - 1], t);
      dwv[M - 1] = t;
    }
    if (isx) {
#pragma unroll
      for (int s = 0; s < M; ++s) Lv[V_DW + s * NX + j] = dwv[s];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    QD_SB();
    // ---- rhs = -(r_w + (Sigma_w + delta) dw + H_ww dw + H_wu du + [end-point slot] d nu), row (s, b) on lane b
    double rhs[M];
    sfor<M>([&](auto S_) {
      constexpr int s = S_;
      double t = fma(sgv[s], dwv[s], rwv[s]);
      if constexpr (s == M - 1) t += dnu;
      if constexpr (s < DEG) {
        // row b of the lambda-weighted Hessian of point s over (x_s, u): its structural non-zeros (packed upper triangle and its mirror)
        sfor<NA>([&](auto I_) {
          constexpr int i = I_;
          sfor<NA - i>([&](auto D_) {
            constexpr int k = i + D_;
            constexpr MoRef rf = moref_dyn(s, MOH_H0 + symi(i, k, NA));
            if constexpr (rf.kind != 0 && i < NX) {
              const double h = mo_dyn_val<rf.kind, rf.off>(rec);
              if constexpr (k < NX) {
                const double pk_ = h * (double)Lv[V_DW + s * NX + k];
                t += (j == i) ? pk_ : 0.0;
                if constexpr (i != k) { const double pi_ = h * (double)Lv[V_DW + s * NX + i]; t += (j == k) ? pi_ : 0.0; }
              } else {
                const double pu_ = h * du[k - NX];
                t += (j == i) ? pu_ : 0.0;
              }
            }
          });
        });
      }
      rhs[s] = -t;
      pin(rhs[s]);
    });
    QD_SB();
    // ---- d lambda = G_w^-T rhs,  G_w^-T = [[Gi', -Gi'E'], [0, I]]: rr = rhs_c + D rhs_e handed round, column (s, b) of Gi times rr
    if (isx) {
#pragma unroll
      for (int s = 0; s < DEG; ++s) Lv[V_RR + s * NX + j] = fma(DOMPC_D[s + 1], rhs[M - 1], rhs[s]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double dl[M];
    if (!fb) {
#pragma unroll
      for (int s = 0; s < DEG; ++s) {
        double t = 0.0;
#pragma unroll
        for (int k2 = 0; k2 < R; ++k2) t = fma(bc[s][k2], (double)Lv[V_RR + k2], t);
        dl[s] = t;
      }
    } else {
#pragma unroll
      for (int s = 0; s < DEG; ++s) {
        double t = 0.0;
        for (int k2 = 0; k2 < R; ++k2) t += ew[EW_LU + k2 * LU_N + s * NX + b] * Lv[V_RR + k2];
        dl[s] = t;
      }
    }
    dl[M - 1] = rhs[M - 1];
    // ---- stores
    if (st_x) {
#pragma unroll
      for (int s = 0; s < M; ++s) {
        Q.dx[pk.woff + s * NX + j] = dwv[s];
        Q.dlam[pk.row0 + s * NX + j] = dl[s];
      }
      if (!chain) Q.dlam[pk.row0 + NW + j] = dnu;
    }
    if constexpr (NE > 0) {
      // nl_cons rows: ds = r_d + J_d dy (- sg deps), d y_d = (Sigma_s + delta) ds + r_s      (row i on lane i)
      if (act && j < NE) {
        const double* S_ = Q.es + (int64_t)e * ES_SIZE;
        double t = S_[ES_RDN + j];
        for (int bb = 0; bb < NA; ++bb) t += ew[EW_JD + j * NA + bb] * Lv[V_DY + bb];
        if (!EPS_GLOBAL && nl_slack(j) >= 0) t -= Q.sgn[e * NE1 + j] * Q.dx[pk.epsoff + nl_slack(j)];
        Q.ds[e * NE1 + j] = t;
        Q.dlam[pk.row0 + NW + NX + j] = (S_[ES_SIGS + j] + delta) * t + S_[ES_RSN + j];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bank ^= 1;
    qd = qn;
  }
}
#elif !defined(DOMPC_HOST_EMU)
__device__ inline void phase_forward_quads(const void*, int, int, int, double, double, int) {}      // (never called: QUAD_FWD is false)
#endif
