// dompc_riccati4.h - backward Riccati recursion over the scenario chains with FOUR CHAINS PER WAVEFRONT (round 6, gfx950 only).
//
// Below the robust horizon every node has one child of the same scenario (optimizer.py:1011-1048): the recursion walks each scenario
// chain from its leaf upwards, one node after the other - the sequential part of an interior-point iteration.  dompc_riccati16.h gives
// every node a whole wavefront (a 16 x 16 problem on 64 lanes, 84 FP64 matrix instructions + 1 800 vector instructions per node; FP64
// matrix instructions have the vector ALU's rate on gfx950) and a wavefront of the batch path walks its problem's nine chains one after
// the other: 171 dependent node updates per pass, 25 % of a solve after the sweep was rewritten (profiles/r06_phase_cycles*.txt).
// Here 16 lanes own one chain and a wavefront walks four chains at once: lane j of a row of 16 owns COLUMN j of every matrix over
// z = (x, u_prev, u) (NYT <= 16 entries):
//     T = P_c F       lane-local: column j of F times the child's value function, whose entries are group-uniform LDS reads
//     F' T            rows of F = rows of [A B] in the staged head of the edge record: group-uniform LDS reads again
//     Cholesky of the NV x NV block Q_vv, gains K, kv: uniform arithmetic per row of 16 (six values fetched with DPP broadcasts)
//     P = Lc' Q_own Lc + Acl' P_c Acl ("Joseph" form, as dompc_riccati16.h): Acl' (P_c Acl) with Acl staged in LDS
// ~1 000 vector instructions and ~800 LDS reads per update of FOUR nodes.  The algebra, the order of the node's own terms and what is
// stored (ND_P, ND_PV, ND_K, ND_KV) are those of r16::node(); the sums are taken in another order (rounding-level differences).
// Models: single child per chain node, no nl_cons rows / slack variables (NE = NS = 0), default rterm.  Everything else - the branching
// levels of the tree included - stays on dompc_riccati16.h.
#pragma once

namespace dompc {
namespace r4 {
#ifndef DOMPC_QUAD_BACKWARD
#define DOMPC_QUAD_BACKWARD 0      // MEASURED, NOT FASTER (round 6): 141 M instead of 111 M cycles of problem 0's wavefront per solve in this pass, the same
#endif                            // MPC steps/s (9 484 vs 9 494, same box, interleaved) - the batch path is bound by memory traffic, not by instruction issue:
                                  // 12 x fewer issue slots per node change nothing.  Kept as the A/B it was measured with (profiles/r06_backward4.txt).
#if !defined(DOMPC_HOST_EMU) && defined(DOMPC_HAVE_QUAD_HELPERS) && DOMPC_NX + 2 * DOMPC_NU <= 16 && DOMPC_NE == 0 && DOMPC_NS == 0
constexpr bool ENABLED = (DOMPC_QUAD_BACKWARD != 0) && R16_ENABLED && (NE == 0) && (NS == 0) && (NV == NU) && (NYT <= 16) && !EPS_GLOBAL;
// LDS of a wavefront (doubles): the staged heads of four edge records [A B | c | Q~ | q~ + r_y], the four children's value functions
// (packed upper triangle + p), and one region for K / v first, the closed-loop maps Acl afterwards
constexpr int ESH = ((ES_QV + NA + 31) / 32) * 32;            // staged doubles per edge record (a multiple of 32: 16 lanes x 16 B per instruction)
constexpr int PT = NA * (NA + 1) / 2;                         // packed P
constexpr int pad4(int n) { return n + ((4 - n % 16) + 16) % 16; }      // smallest size >= n that is 4 mod 16: the four chains' copies then start
                                                                       // in different LDS banks (group-uniform 8-byte reads of four addresses: no conflict)
constexpr int PG = pad4(PT + NA);                             // ... + p, per chain
constexpr int AG = pad4(NA * NA > 64 ? NA * NA : 64);         // Acl, per chain (K and the vector v live here before)
constexpr int EG = pad4(ESH);                                 // staged head of an edge record, per chain
constexpr int L_ES = 0, L_P = 4 * EG, L_A = L_P + 4 * PG, L_C = L_A, L_END = L_A + 4 * AG;      // L_C: (dg, gv) of the next node, per lane - inside the Acl region: written at the end of an update (Acl is dead), read at the top of the next one (before K is written)
static_assert(!ENABLED || L_END <= EL_SIZE, "working set of the four-chain recursion must fit the wavefront's LDS region");
static_assert(!ENABLED || 3 * 16 + 16 <= AG, "K and v share the Acl region");
constexpr int RB = 3;                                         // rows per batch of group-uniform LDS reads in the big products
constexpr int psym(int i, int k) { return i <= k ? i * NA - i * (i - 1) / 2 + k - i : k * NA - k * (k - 1) / 2 + i - k; }
// index of z-entry i inside y = (x_n, u_n) of the condensed edge blocks, or -1 (u_prev)
constexpr int yz(int i) { return (i < NX) ? i : ((i >= NA && i < NA + NU) ? NX + (i - NA) : -1); }

struct Val4 { double P[NA]; double p; };                     // lane j < NA: column j of P and p_j

// heads of the edge records of the four chains -> LDS.  Requested into registers in the middle of a node update (16 doubles per lane:
// lane jl of a row of 16 takes the 16-byte pieces jl, jl + 16, ...) and written to the chain's copy at the end of the update: an LDS-DMA
// copy lands at a lane-determined address - the four copies 32 doubles apart, every group-uniform read of the four a 4-way bank conflict
// (measured: the LDS of a CU, shared by eight wavefronts, became the bottleneck: 46 k cycles per update of four nodes).
typedef double d2_ __attribute__((ext_vector_type(2)));
struct Pre4 { d2_ v[ESH / 32]; };
__device__ inline void fetch4(const Prob& Q, int e_lane, int jl, Pre4& R) {
  const d2_* src = (const d2_*)(Q.es + (int64_t)e_lane * ES_SIZE) + jl;
#pragma unroll
  for (int q = 0; q < ESH / 32; ++q) R.v[q] = src[16 * q];
}
__device__ inline void put4(ldsd* Les, int jl, const Pre4& R) {
  typedef __attribute__((address_space(3))) d2_ lds_d2;
  lds_d2* dst = (lds_d2*)Les + jl;
#pragma unroll
  for (int q = 0; q < ESH / 32; ++q) dst[16 * q] = R.v[q];
}
__device__ inline constexpr int es_at(int i) { return i; }      // entry i of the staged record of this lane's chain
// the lane number, computed HERE (volatile: neither hoisted out of a loop nor kept - spilled - across it)
__device__ inline int lane_id_here() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// Table entries of the chain nodes n0 .. n0 + 3 (one per row of 16 lanes; rows beyond `cnt` repeat the last node): read with the UNIFORM
// index n0 + r - scalar loads through the constant cache - and selected per row.  Indexed with the lane's own node number every one of
// them is a vector load, and the chains of dependent look-ups (node -> in-edge -> row offset, node -> parent -> its input offset,
// node -> child edge -> its weight) were 30 k cycles per update (profiles/r06_backward4_sections.txt).
struct NodeIdx { int xo, uo, ie, row0_in, up_off, e_child; double rw; };
__device__ inline NodeIdx node_idx4(const KArgs& A, const Prob& Q, int n0, int cnt, int g) {
  NodeIdx R{0, 0, -1, 0, -1, 0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + (r < cnt ? r : cnt - 1);                 // uniform
    const int xo = A.node_x_off[n], uo = A.node_u_off[n], ie = A.node_in_edge[n], pn = A.node_parent[n];
    const int cs = A.node_child_start[n], cc = A.node_child_count[n];
    const int row0 = ie >= 0 ? A.edge_row0[ie] : 0;
    const int upo = pn >= 0 ? A.node_u_off[pn] : -1;
    const double rw = cc > 0 ? cc * A.edge_omega[cs] * Q.sf : 0.0;
    const bool me = (g == r);
    R.xo = me ? xo : R.xo; R.uo = me ? uo : R.uo; R.ie = me ? ie : R.ie; R.row0_in = me ? row0 : R.row0_in;
    R.up_off = me ? upo : R.up_off; R.e_child = me ? cs : R.e_child; R.rw = me ? rw : R.rw;
  }
  return R;
}
// the node's variable data for z-entry j (r16::load_node with the indices above)
__device__ inline void load_node4(const KArgs& A, const Prob& Q, const NodeIdx& I, int j, r16::NodeIn& R) {
  const bool is_up = (j >= NX && j < NA);
  const int jj = j < NYT ? j : 0;
  const int gi = (jj < NX) ? I.xo + jj : (is_up ? I.uo + (jj - NX) : I.uo + (jj - NA));
  R.xv = Q.x[gi]; R.lo = Q.lb[gi]; R.hi = Q.ub[gi]; R.zlo = Q.zl[gi]; R.zhi = Q.zu[gi];
  const int iu = is_up ? jj - NX : (jj >= NA ? jj - NA : 0);
  R.upv = (jj >= NX) ? (I.up_off >= 0 ? Q.x[I.up_off + iu] : Q.P[A.p_off_uprev + iu] / tab_sel(DOMPC_SU, iu)) : 0.0;
  R.nu = (jj < NX) ? ((I.ie >= 0) ? Q.lam[I.row0_in + NW + jj] : Q.lam[jj]) : 0.0;
}

// Walk the scenario chains of the problem from their leaves up to level `cl`, four chains per wavefront.  Returns 1 if a Q_vv block
// was not positive definite (the caller sets the failure flag).
__device__ __attribute__((noinline)) int phase_chains(const void* kp, int b_, int slot, int soc, double sf, double mu_, double delta_, int cl_) {
  // (its own function, like the quad loop of the sweep: its own register allocation, nothing of the caller's live across the walk)
  const KArgs A = kernel_args(kp);
  Thr T = make_thr(A);
  T.kp = kp;
  Prob Q = make_prob(A, ufl(slot), A.p + (int64_t)ufl(b_) * A.n_opt_p);
  Q.sf = ufl(sf);
  Q.soc = ufl(soc);
  prob_bounds(Q);
  const double mu = ufl(mu_), delta = ufl(delta_);
  const int cl = ufl(cl_);
  const int ng = T.nt / 64, gid = group_index(T.tid, 64);
  ldsd* Ld = T.edge_lds + (int64_t)(T.ltid / 64) * EL_SIZE;
  const int S = A.level_node_start[A.N + 1] - A.level_node_start[A.N];
  const int nq = (S + 3) / 4;
  const bool damp = !(Q.soc & 2);
  int bad = 0;
#if DOMPC_PROFILE
  long long pc0 = prof_clock();
#define R4_PN(i) if (T.prof && T.tid == 0) { const long long pc1 = prof_clock(); T.prof[i] += pc1 - pc0; pc0 = pc1; }
#else
#define R4_PN(i)
#endif
  // own terms of z-entry j of node n (r16::node: diagonal dg, gradient gv without the edge's q~ + r_y), from the node's variable data
  auto own_terms = [&](const r16::NodeIn& in, const NodeIdx& I, int j, double& dg, double& gv) {
    const bool is_x = j < NX, is_up = (j >= NX && j < NA), is_u = (j >= NA && j < NA + NU);
    const int iu = is_up ? j - NX : (is_u ? j - NA : 0);
    const double rt_j = (is_up || is_u) ? tab_sel(DOMPC_RTERM, iu) : 0.0;
    const int ie = I.ie;
    const double rw = I.rw, rwh = (Q.soc & 2) ? 0.0 : rw;
    const double xv = in.xv, lo = in.lo, hi = in.hi, upv = in.upv;
    if (is_up) {
      dg = 2.0 * rwh * rt_j;
      gv = -2.0 * rw * rt_j * (xv - upv);
    } else {
      dg = sigma_of(xv, lo, hi, in.zlo, in.zhi) + delta;
      gv = bar_grad(xv, lo, hi, mu, damp);
      if (is_x) gv += (ie >= 0) ? -in.nu : in.nu;
      else { dg += 2.0 * rwh * rt_j; gv += 2.0 * rw * rt_j * (xv - upv); }
    }
    if (j >= NYT) { dg = 0.0; gv = 0.0; }
  };
  for (int qd = gid; qd < nq; qd += ng) {
    // (the lane number as a value the optimiser cannot see through, dompc_quad.h: what is derived from it is recomputed where it is used
    //  instead of being held - or spilled - across the walk)
    const int lane = lane_id_here();
    const int g = lane >> 4, j = lane & 15;
    const bool is_x = j < NX;
    const int s_raw = 4 * qd + g;
    const bool act = s_raw < S;
    const int s_ = act ? s_raw : S - 1;                         // (rows beyond the last chain repeat it and store nothing)
    // ---- leaf: P = sf*omega*Hm + Sigma_x + delta over x, p = sf*omega*gm - nu_in + barrier gradient
    Val4 V;
    {
      const int n = A.level_node_start[A.N] + s_;
      const NodeIdx IL = node_idx4(A, Q, A.level_node_start[A.N] + 4 * qd, (S - 4 * qd < 4) ? S - 4 * qd : 4, g);
      const int ie = IL.ie;
      const double* S_ = Q.es + (int64_t)ie * ES_SIZE;
      const int xo = IL.xo;
      const int jj = is_x ? j : 0;
      const double xv = Q.x[xo + jj], lo = Q.lb[xo + jj], hi = Q.ub[xo + jj];
      const double dgl = sigma_of(xv, lo, hi, Q.zl[xo + jj], Q.zu[xo + jj]) + delta;
      const double gvl = S_[ES_MG + jj] - Q.lam[IL.row0_in + NW + jj] + bar_grad(xv, lo, hi, mu, damp);
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        double v = (i < NX) ? S_[ES_MH + i * NX + jj] : 0.0;
        if (i == j) v += dgl;
        V.P[i] = (is_x && i < NX) ? v : 0.0;
      }
      V.p = is_x ? gvl : 0.0;
      if (act && j < NA) {
        double* Nd = Q.nd + (int64_t)n * ND_SIZE;
#pragma unroll
        for (int i = 0; i < NA; ++i) Nd[ND_P + i * NA + j] = V.P[i];
        Nd[ND_PV + j] = V.p;
      }
    }
    // ---- the chain, node (k, s) -> parent (k - 1, s)
    // Only (dg, gv) of the NEXT node travel from one update to the next: its variable data are requested in the middle of an update
    // and turned into these two values at its end - seven loop-carried values per lane went through scratch, and a scratch reload behind
    // the node's stores waits for them to reach memory (dompc_quad.h)
    double dg_c = 0.0, gv_c = 0.0;
    const int cnt = (S - 4 * qd < 4) ? S - 4 * qd : 4;          // chains of this quad (uniform)
    if (A.N - 1 >= cl) {
      const NodeIdx I0 = node_idx4(A, Q, A.level_node_start[A.N - 1] + 4 * qd, cnt, g);
      r16::NodeIn in0;
      load_node4(A, Q, I0, j, in0);
      Pre4 pre0;
      fetch4(Q, I0.e_child, j, pre0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      own_terms(in0, I0, j, dg_c, gv_c);
      put4(Ld + L_ES + g * EG, j, pre0);
      Ld[L_C + lane] = dg_c; Ld[L_C + 64 + lane] = gv_c; Ld[L_C + 128 + lane] = I0.rw;
    }
    for (int k = A.N - 1; k >= cl; --k) {
      const int lane = lane_id_here();
      const int g = lane >> 4, j = lane & 15;
      const bool act = 4 * qd + g < S;
      ldsd* Les = Ld + L_ES + g * EG;
      ldsd* Lp = Ld + L_P + g * PG;
      ldsd* La = Ld + L_A + g * AG;
      const int yj = (j < NX) ? j : ((j >= NA && j < NA + NU) ? NX + (j - NA) : -1);
      const int n = A.level_node_start[k] + 4 * qd + (g < cnt ? g : cnt - 1);      // (no table look-up with a per-row index anywhere in the update)
      // the child's value function -> LDS (packed upper triangle: lane j holds entries (i, j), i <= j; p behind it)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (j < NA) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
          if (i <= j) Lp[i * NA - i * (i - 1) / 2 + j - i] = V.P[i];
        Lp[PT + j] = V.p;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the staged heads have landed: waited for at the end of the previous update)
      __builtin_amdgcn_wave_barrier();
      QD_SB();
      // ---- own terms of z-entry j (computed when the node's data arrived) + the edge's q~ + r_y
      double dg = Ld[L_C + lane], gv = Ld[L_C + 64 + lane];      // (through LDS, not through loop-carried registers: those ended up in scratch)
      const double rw = Ld[L_C + 128 + lane], rwh = (Q.soc & 2) ? 0.0 : rw;
      { const double t_ = Les[es_at(ES_QV + (yj >= 0 ? yj : 0))]; gv += (yj >= 0 && j < NYT) ? t_ : 0.0; }
      // column j of F over the child's state x+ (rows k < NX): A over x, 0 over u_prev, B over u; rows NX + u: unit vectors of the u columns
      double fk[NX];
#pragma unroll
      for (int kk = 0; kk < NX; ++kk) { const double t_ = Les[es_at(ES_AB + kk * NA + (yj >= 0 ? yj : 0))]; fk[kk] = (yj >= 0) ? t_ : 0.0; }
      // column j of Q_own: Q~ over (x, u), the diagonal, the rterm coupling of u_prev and u
      double Qo[NYT];
      sfor<NYT>([&](auto I_) {
        constexpr int i = I_, yi = yz(i);
        double v = 0.0;
        if constexpr (yi >= 0) {
          const int yjc = yj >= 0 ? yj : 0;
          const int ii = ES_QT + ((yi <= yjc) ? (yi * NA - yi * (yi - 1) / 2 + yjc - yi) : (yjc * NA - yjc * (yjc - 1) / 2 + yi - yjc));
          const double t_ = Les[ii];
          v = (yj >= 0) ? t_ : 0.0;
        }
        if (i == j) v += dg;
        if constexpr (i >= NX && i < NA) v -= (j == i + NU) ? 2.0 * rwh * DOMPC_RTERM[i - NX] : 0.0;      // row u_prev_i, column u_i
        if constexpr (i >= NA && i < NA + NU) v -= (j == i - NU) ? 2.0 * rwh * DOMPC_RTERM[i - NA] : 0.0;      // row u_i, column u_prev_i
        Qo[i] = v;
      });
#pragma unroll
      for (int i = 0; i < NYT; ++i) pin(Qo[i]);
      QD_SB();
      R4_PN(12)
      // ---- v = P_c c + p_c (lane i: entry i, from its own column = row of the symmetric P_c), handed round through LDS
      {
        double t = V.p;
#pragma unroll
        for (int m = 0; m < NX; ++m) t = fma(V.P[m], (double)Les[es_at(ES_CV + m)], t);
        if (j < NA) La[48 + j] = t;
      }
      // ---- T = P_c F (column j): rows i < NA
      double Tc[NA];
      sfor<NA>([&](auto I_) {
        constexpr int i = I_;
        if constexpr (i % RB == 0) QD_SB();          // (a few rows per batch of LDS reads: the scheduler would otherwise issue all of them up front and spill)
        double t = 0.0;
        sfor<NX>([&](auto K_) { constexpr int kk = K_; t = fma((double)Lp[psym(i, kk)], fk[kk], t); });
        sfor<NU>([&](auto U_) { constexpr int u = U_; const double pv = Lp[psym(i, NX + u)]; t += (j == NA + u) ? pv : 0.0; });
        Tc[i] = t;
        pin(Tc[i]);
      });
      QD_SB();
      R4_PN(13)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // ---- Q_tot (column j) = Q_own + F' T ; q_tot (entry j) = gv + F' v
      // (only the rows of the decision variables are needed: the value function is rebuilt in closed-loop form below)
      double Qt[NYT], qt = gv;
      sfor<NV>([&](auto U_) {
        constexpr int i = NA + U_, yi = yz(i);
        double t = Qo[i];
        sfor<NX>([&](auto K_) { constexpr int kk = K_; t = fma((double)Les[es_at(ES_AB + kk * NA + yi)], Tc[kk], t); });
        t += Tc[NX + (i - NA)];
        Qt[i] = t;
        pin(Qt[i]);
      });
      QD_SB();
      {
        double t = 0.0;
#pragma unroll
        for (int kk = 0; kk < NX; ++kk) t = fma(fk[kk], (double)La[48 + kk], t);
        sfor<NU>([&](auto U_) { constexpr int u = U_; const double vv = La[48 + NX + u]; t += (j == NA + u) ? vv : 0.0; });
        qt += t;
      }
      pin(qt);
      QD_SB();
      R4_PN(14)
      // ---- Cholesky of Q_vv (uniform per row of 16), gains for this lane's column
      double kv[NV], Kj[NV];
      {
        double qvv[NV * NV], qv[NV], qx[NV], L[NV * NV], Li[NV];
        sfor<NV>([&](auto U_) {
          constexpr int u = U_;
          sfor<u + 1>([&](auto W_) { constexpr int w = W_; qvv[u * NV + w] = rbc<NA + w>(Qt[NA + u]); });
          qv[u] = rbc<NA + u>(qt);
          qx[u] = Qt[NA + u];
        });
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
        auto solve = [&](double* y) {
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
      // K -> LDS (row u at [16 u + j]) for the group-uniform reads of the products with Lc = [I; K]
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int u = 0; u < NV; ++u) La[16 * u + j] = Kj[u];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      QD_SB();
      R4_PN(10)
      // ---- own part: column j of Lc' Q_own Lc, entry j of Lc'(q_own + Q_own l0)
      double Pn[NA], pn;
      {
        double dgu[NV];
        sfor<NV>([&](auto U_) { constexpr int u = U_; dgu[u] = rbc<NA + u>(dg); });
        // U = Q_own Lc (column j): Q_own[:, j] + sum_u Q_own[:, NA + u] K[u][j]
        double Uo[NYT];
        sfor<NYT>([&](auto I_) {
          constexpr int i = I_, yi = yz(i);
          if constexpr (i % (2 * RB) == 0) QD_SB();
          double t = Qo[i];
          sfor<NV>([&](auto U_) {
            constexpr int u = U_;
            double qiu = 0.0;                               // Q_own[i][NA + u]: group-uniform
            if constexpr (yi >= 0) qiu = Les[es_at(ES_QT + psym(yi, NX + u))];
            if constexpr (i == NA + u) qiu += dgu[u];
            if constexpr (i == NX + u) qiu -= 2.0 * rwh * DOMPC_RTERM[u];
            t = fma(qiu, Kj[u], t);
          });
          Uo[i] = t;
        });
        sfor<NA>([&](auto I_) {
          constexpr int i = I_;
          double t = Uo[i];
          sfor<NV>([&](auto U_) { constexpr int u = U_; t = fma((double)La[16 * u + i], Uo[NA + u], t); });
          Pn[i] = t;
        });
        // w = q_own + Q_own l0, l0 = (0; kv): entry j = gv + sum_u Q_own[NA + u][j] kv[u] (the symmetric column); then Lc' w
        double wj = gv;
#pragma unroll
        for (int u = 0; u < NV; ++u) wj = fma(Qo[NA + u], kv[u], wj);
        double t = wj;
        sfor<NV>([&](auto U_) { constexpr int u = U_; t = fma(Kj[u], rbc<NA + u>(wj), t); });
        pn = t;
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) pin(Pn[i]);
      pin(pn);
      QD_SB();
      R4_PN(8)
      // ---- closed loop: Acl = F Lc (column j < NA), ccl = F l0 + c (uniform), T2 = P_c Acl, t2v = P_c ccl + p_c
      double Ac[NA];
#pragma unroll
      for (int kk = 0; kk < NX; ++kk) {
        double t = (j < NA) ? fk[kk] : 0.0;
        sfor<NV>([&](auto U_) { constexpr int u = U_; t = fma((double)Les[es_at(ES_AB + kk * NA + NX + u)], Kj[u], t); });
        Ac[kk] = t;
      }
#pragma unroll
      for (int u = 0; u < NV; ++u) Ac[NX + u] = Kj[u];
      double t2v;                                            // entry j of P_c ccl + p_c (lane j: its own column = row of P_c)
      {
        double t = V.p;
#pragma unroll
        for (int m = 0; m < NX; ++m) {
          double cc_ = Les[es_at(ES_CV + m)];
          sfor<NV>([&](auto U_) { constexpr int u = U_; cc_ = fma((double)Les[es_at(ES_AB + m * NA + NX + u)], kv[u], cc_); });
          t = fma(V.P[m], cc_, t);
        }
#pragma unroll
        for (int u = 0; u < NV; ++u) t = fma(V.P[NX + u], kv[u], t);
        t2v = t;
      }
      // (the staged heads are consumed: the parent's operands - its variable data, the head of its child edge - are requested now and
      //  arrive during the two products below)
      Pre4 pre;
#pragma unroll
      for (int q = 0; q < ESH / 32; ++q) pre.v[q] = d2_{0.0, 0.0};
      r16::NodeIn nx;
      nx.xv = nx.lo = nx.hi = nx.zlo = nx.zhi = nx.nu = nx.upv = 0.0;
      NodeIdx Ip{0, 0, -1, 0, -1, 0, 0.0};
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (k > cl) {
        Ip = node_idx4(A, Q, A.level_node_start[k - 1] + 4 * qd, cnt, g);
        load_node4(A, Q, Ip, j, nx);
        fetch4(Q, Ip.e_child, j, pre);
      }
      double T2[NA];
      sfor<NA>([&](auto I_) {
        constexpr int i = I_;
        if constexpr (i % RB == 0) QD_SB();
        double t = 0.0;
        sfor<NA>([&](auto K_) { constexpr int kk = K_; t = fma((double)Lp[psym(i, kk)], Ac[kk], t); });
        T2[i] = t;
        pin(T2[i]);
      });
      QD_SB();
      R4_PN(9)
      // Acl and t2v -> LDS (K and v are dead), then P += Acl' T2 , p += Acl' t2v
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (j < NA) {
#pragma unroll
        for (int kk = 0; kk < NA; ++kk) La[kk * NA + j] = Ac[kk];
      }
      const double t2b = t2v;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      QD_SB();
      sfor<NA>([&](auto I_) {
        constexpr int i = I_;
        if constexpr (i % RB == 0) QD_SB();
        double t = Pn[i];
        sfor<NA>([&](auto K_) { constexpr int kk = K_; t = fma((double)La[kk * NA + i], T2[kk], t); });
        V.P[i] = t;
        pin(V.P[i]);
      });
      QD_SB();
      {
        double t = pn;
        sfor<NA>([&](auto K_) { constexpr int kk = K_; t = fma(Ac[kk], rbc<kk>(t2b), t); });
        V.p = t;
      }
      if (j >= NA) {
#pragma unroll
        for (int i = 0; i < NA; ++i) V.P[i] = 0.0;
        V.p = 0.0;
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) pin(V.P[i]);
      pin(V.p);
      QD_SB();
      R4_PN(11)
      // ---- the parent's operands have landed (nothing else is in flight): its own terms; then the stores of this node at its very
      //      end (r16::node) - they reach memory during the parent's update
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (k > cl) {
        own_terms(nx, Ip, j, dg_c, gv_c);
        Ld[L_C + lane] = dg_c; Ld[L_C + 64 + lane] = gv_c; Ld[L_C + 128 + lane] = Ip.rw;
        put4(Les, j, pre);                            // (every read of this node's heads is through: fences above)
      }
      if (act && j < NA) {
        double* Nd = Q.nd + (int64_t)n * ND_SIZE;
#pragma unroll
        for (int u = 0; u < NV; ++u) Nd[ND_K + u * NA + j] = Kj[u];
#pragma unroll
        for (int i = 0; i < NA; ++i) Nd[ND_P + i * NA + j] = V.P[i];
        Nd[ND_PV + j] = V.p;
        if (j == 0) {
#pragma unroll
          for (int u = 0; u < NV; ++u) Nd[ND_KV + u] = kv[u];
        }
      }
      R4_PN(15)
      if (__ballot(bad) != 0ull) break;
    }
    if (__ballot(bad) != 0ull) break;
  }
  return __ballot(bad) != 0ull;
}
__device__ inline int chains(const Thr& T, const Prob& Q, double mu, double delta, int cl) {
  const KArgs& A = *Q.A;
  return phase_chains(T.kp, (int)((Q.P - A.p) / A.n_opt_p), Q.slot, Q.soc, Q.sf, mu, delta, cl);
}
#else
constexpr bool ENABLED = false;
#ifndef DOMPC_HOST_EMU
__device__ inline int chains(const Thr&, const Prob&, double, double, int) { return 0; }      // (never called)
#endif
#endif
}  // namespace r4
}  // namespace dompc
