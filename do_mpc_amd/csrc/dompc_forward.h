// dompc_forward.h - structured interior-point solver, part of dompc_kernel.h (included there, inside namespace dompc, in this order:
// dompc_edge.h, dompc_factor.h, dompc_node.h, dompc_riccati.h, dompc_forward.h, dompc_sweep.h, dompc_phases.h, dompc_driver.h).
// Contents: forward pass: node steps (chain walk), per-edge collocation and multiplier steps, adjoint recovery of the continuity multipliers.
// Sizes, record layouts, the thread context `Thr`, reductions and the small dense products are in dompc_kernel.h.


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
  // root.  When the tree branches at the root (chain_level >= 1) the wavefront that takes node 0 forms the root's step itself, inside
  // node_step: one device-scope barrier less per pass.  Otherwise thread 0 does it, followed by a barrier, as before.
  const int cl = A.chain_level < A.N ? A.chain_level : A.N;
#ifndef DOMPC_HOST_EMU
  const bool root_fused = cl >= 1 && !SHARD;
#else
  const bool root_fused = false;
#endif
  if (!root_fused) {
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
  }
  // node steps: dv = K dx~ + kv, children dx~ = Atilde [dx~; dv] + c~.  Branching stages level by level with
  // a barrier; below the robust horizon each group walks its scenario chain downwards with dx~ kept in LDS.
  auto node_step = [&](int n) {                         // generic (any number of children; operands from global memory)
    const double* Nd = Q.ND(n);
    if (root_fused && n == 0) {
      const int xo = A.node_x_off[0];
      for (int a = lane; a < NA; a += GS) {
        const double t = a < NX ? (FREE_ROOT ? Nd[ND_DXT + a] : -Q.c[a]) : 0.0;
        Ld[RF_DX + a] = t;
        if (!FREE_ROOT || a >= NX) Q.ND(0)[ND_DXT + a] = t;
        if (a < NX) Q.dx[xo + a] = t;
      }
    } else
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
    // One chain per wavefront (one problem alone on the chip, large trees): the walk is then a chain of memory round trips - a step's
    // operands, requested one step ahead, take 4 k cycles to arrive while the step itself takes 600.  The three idle lane groups are put
    // to use as PREFETCH DEPTH: lane group q requests and holds the operands of the steps k = cl + q (mod 4), four steps are in flight,
    // and the steps are executed one after the other by "their" lane group on one shared set of step vectors (same arithmetic, same bits).
#ifndef DOMPC_FW_DEEP
#define DOMPC_FW_DEEP 1
#endif
    const bool deep = DOMPC_FW_DEEP && cw == 1;
    const int kstep = deep ? 4 : 1;
    for (int s0 = cw * gid; s0 < S && cl < A.N; s0 += cw * ng) {
      const bool here = deep || (c4 < cw && s0 + c4 < S);
      const int sc = deep ? s0 : (here ? s0 + c4 : S - 1);   // (lane groups without a chain repeat the last one and store nothing)
#ifdef DOMPC_FW_NOSTORE            /* timing experiment only: the chain walk without its global stores (wrong steps) */
      const bool on_chain = false;
#else
      const bool on_chain = here && mk_n(A, A.level_node_start[A.N] + sc);
#endif
      ldsd *DXs = deep ? Ld : DX, *DVs = deep ? Ld + 16 : DV, *DXNs = deep ? Ld + 32 : DXN;      // step vectors: the chain's (deep: one set for the wavefront)
      double v[FW4_PL];
      // Indices of a step from SCALAR table reads: the four lane groups of a wavefront work on four (level, chain) pairs that are uniform
      // per group - (k + q, s0) in deep mode, (k, s0 + q) otherwise - so each group's node / edge numbers and offsets are read with scalar
      // loads and picked by lane group.  (They used to be per-lane vector reads: a chain of three dependent memory round trips in front of
      // the operand requests, and vector loads whose results - carried into the next step through register copies - made every step wait
      // for ALL the requests it had just issued: the prefetch overlapped with nothing, 4 k cycles per step of a 600-cycle computation.)
      auto load4 = [&](int kb, Ix& ix) {
        int n_[4], e_[4], cn_[4], uo_[4], eo_[4], xoc_[4], row0_[4];      // (uniform: scalar registers; the reads of one stage side by side)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kq_ = deep ? kb + q : kb;
          const int kq = kq_ < A.N ? kq_ : A.N - 1;          // (deep: lane groups beyond the last step repeat it and store nothing)
          const int scq = deep ? s0 : ((q < cw && s0 + q < S) ? s0 + q : S - 1);
          n_[q] = A.level_node_start[kq];
          cn_[q] = A.level_node_start[kq + 1] + scq;
          e_[q] = scq;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { e_[q] += A.node_child_start[n_[q]]; n_[q] += deep ? s0 : ((q < cw && s0 + q < S) ? s0 + q : S - 1); }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uo_[q] = A.node_u_off[n_[q]]; eo_[q] = NS > 0 ? A.node_eps_off[n_[q]] : 0; xoc_[q] = A.node_x_off[cn_[q]]; row0_[q] = A.edge_row0[e_[q]];
        }
        auto pick = [&](const int (&a)[4]) { return c4 == 0 ? a[0] : (c4 == 1 ? a[1] : (c4 == 2 ? a[2] : a[3])); };
        const unsigned nd0 = (unsigned)pick(n_) * (unsigned)ND_SIZE, es0 = (unsigned)pick(e_) * (unsigned)ES_SIZE, nc0 = (unsigned)pick(cn_) * (unsigned)ND_SIZE;
        const int uo = pick(uo_), eo = pick(eo_), xoc = pick(xoc_), row0 = pick(row0_);
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
        ix.uo = uo; ix.eo = eo; ix.xoc = xoc; ix.row0 = row0; ix.ndc = nc0;
      };
      Ix cur, nxt;
      load4(cl, cur);
      if (ll < NA && (!deep || c4 == 0)) DXs[ll] = ldoff(Q.nd, (unsigned)(A.level_node_start[cl] + sc) * (unsigned)ND_SIZE + (unsigned)(ND_DXT + ll));
      for (int k = cl; k < A.N; k += kstep) {
#pragma unroll
        for (int q = 0; q < FW4_PL; ++q) {
          const int i = ll + 16 * q;
          if (i < FW_N) IN[i] = v[q];
        }
        if (k + kstep < A.N) load4(k + kstep, nxt);      // (in flight during the step(s))
        T.gsync();
        const ldsd *K_ = IN, *KV_ = K_ + NV * NA, *AB_ = IN + FW_K, *CV_ = AB_ + NX * NA, *PC_ = IN + FW_K + FW_AB, *PV_ = PC_ + NX * NA;
        for (int qd = 0; qd < kstep; ++qd) {
          const bool act = !deep || (c4 == qd && k + qd < A.N);       // this lane group's step
          const bool on = on_chain && act;
          if (act && ll < NV) {
            double t = KV_[ll];
#pragma unroll
            for (int a = 0; a < NA; ++a) t += K_[ll * NA + a] * DXs[a];
            DVs[ll] = t;
            if (on) {
              if (ll < NU) Q.dx[cur.uo + ll] = t;
              else Q.dx[cur.eo + ll - NU] = t;
            }
          }
          T.gsync();
          if (act && ll < NA) {
            double t;
            if (ll < NX) {
              t = CV_[ll];
#pragma unroll
              for (int b = 0; b < NX; ++b) t += AB_[ll * NA + b] * DXs[b];
#pragma unroll
              for (int b = 0; b < NU; ++b) t += AB_[ll * NA + NX + b] * DVs[b];
              if (on) Q.dx[cur.xoc + ll] = t;
            } else {
              t = DVs[ll - NX];
            }
            if (on) Q.nd[cur.ndc + (unsigned)(ND_DXT + ll)] = t;
            DXNs[ll] = t;
          }
          T.gsync();
          if (act && ll < NX) {
            double t = PV_[ll];
#pragma unroll
            for (int b = 0; b < NA; ++b) t += PC_[ll * NA + b] * DXNs[b];
            if (on) Q.dlam[cur.row0 + NW + ll] = t;
          }
          if (act && ll < NA) DXs[ll] = DXNs[ll];
          T.gsync();
        }
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
  if (QUAD_FWD && GS == 64 && !adj) {
    // four edges per wavefront, G_cc^-1 formed again from the compact model-output record instead of read from the forward record
    // (dompc_quad.h; its own function: its own register allocation)
    phase_forward_quads(T.kp, (int)((Q.P - A.p) / A.n_opt_p), Q.slot, Q.soc, Q.sf, delta, cl);
  } else if (FE2 && GS == 64 && !adj) {
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

