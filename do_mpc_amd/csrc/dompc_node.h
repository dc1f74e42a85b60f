// dompc_node.h - structured interior-point solver, part of dompc_kernel.h (included there, inside namespace dompc, in this order:
// dompc_edge.h, dompc_factor.h, dompc_node.h, dompc_riccati.h, dompc_forward.h, dompc_sweep.h, dompc_phases.h, dompc_driver.h).
// Contents: gradient / dual-residual assembly of the variables owned by a node.
// Sizes, record layouts, the thread context `Thr`, reductions and the small dense products are in dompc_kernel.h.

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
  // Every operand is loaded BEFORE the first result is stored (round 6).  The loop used to alternate per variable "store gf, load z_L / z_U,
  // store rd": the compiler cannot move a load over a store that may alias, so every s_waitcnt of a load also waited for the store in
  // front of it to reach memory - two to three write round trips per variable, 200 k cycles per trip of this thread-per-node loop and
  // 7 % of a solve (profiles/r06_phase_cycles*.txt: "sweep:node assembly").  Same arithmetic, same order: the same bits.
  double gxv[NX], rxv[NX], zlx[NX], zux[NX];
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
    gxv[a] = gx; rxv[a] = rx;
    zlx[a] = Q.zl[xo + a]; zux[a] = Q.zu[xo + a];
  }
  const int uo = cc > 0 ? A.node_u_off[n] : 0;
  double guv[NU > 0 ? NU : 1], ruv[NU > 0 ? NU : 1];
  if (cc > 0) {
    double tmp[NU];
    const double* up = uprev_ptr(Q, n, Q.x, tmp);
    const double rw = node_rweight(Q, n);
    for (int i = 0; i < NU; ++i) {
      const double rt = (RT_CUSTOM ? 0.0 : 2.0 * rw * DOMPC_RTERM[i] * (Q.x[uo + i] - up[i])) + in[2 * NX + 2 * NU + i];    // (user-defined rterm: own share is in the edges' GFY / RY)
      guv[i] = in[2 * NX + i] + rt;
      ruv[i] = in[2 * NX + NU + i] + rt - Q.zl[uo + i] + Q.zu[uo + i];
    }
  }
  const int eo = (NS > 0 && cc > 0) ? A.node_eps_off[n] : 0;
  double gev[NS1], rev[NS1];
  if (NS > 0 && cc > 0)
    for (int q = 0; q < NS; ++q) {
      gev[q] = cc * Q.sf * DOMPC_EPS_PEN[q];
      rev[q] = gev[q] + in[2 * NX + 3 * NU + q] - Q.zl[eo + q] + Q.zu[eo + q];
    }
  for (int a = 0; a < NX; ++a) {
    Q.gf[xo + a] = gxv[a];
    Q.rd[xo + a] = rxv[a] - zlx[a] + zux[a];
  }
  if (cc == 0) return;
  for (int i = 0; i < NU; ++i) {
    Q.gf[uo + i] = guv[i];
    Q.rd[uo + i] = ruv[i];
  }
  if (NS > 0)
    for (int q = 0; q < NS; ++q) {
      Q.gf[eo + q] = gev[q];
      Q.rd[eo + q] = rev[q];
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

