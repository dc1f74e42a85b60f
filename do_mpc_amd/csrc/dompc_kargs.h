// dompc_kargs.h - kernel argument block shared by the generic host runtime (dompc_runtime.cpp)
// and the per-model device code (dompc_kernel.h).  Plain data, no model-dependent sizes.
#pragma once
#include <stdint.h>
#include "../../include/dompc_ipm.h"

namespace dompc {
struct KArgs {
  const int32_t *level_node_start, *node_level, *node_x_off, *node_u_off, *node_eps_off;
  const int32_t *node_child_start, *node_child_count, *node_parent, *node_in_edge;
  const int32_t *edge_parent, *edge_child, *edge_pidx, *edge_w_off, *edge_row0, *edge_level;
  const double* edge_omega;
  const int32_t* dummy_idx;
  int32_t N, n_nodes, n_edges, n_dummy, n_opt_x, n_opt_p, n_g, e_pad;
  int32_t p_off_tvp, p_off_p, p_off_uprev;
  int32_t chain_level;      // first stage from which every node has exactly one child of the same scenario index (= n_robust)
  // batch I/O (device pointers)
  const double *x0, *lbx, *ubx, *lbg, *ubg, *p;
  double *x_out, *g_out, *lam_x_out, *lam_g_out, *f_out;
  dompc_stats* stats;
  int32_t batch, n_slots;
  double* ws;
  int64_t ws_stride;
  int32_t* work_counter;
  dompc_options opt;
  // debug (mode 1): one Newton step at the given point
  int32_t mode;
  const double *dbg_lam, *dbg_zl, *dbg_zu;
  double dbg_mu, dbg_delta;
  double *dbg_dx, *dbg_dlam, *dbg_rd, *dbg_c;
  // optional iteration trace of problem 0: 8 doubles per iteration (it, mu, E0, inf_pr, inf_du, alpha, delta_w, obj)
  double* trace;
  int32_t trace_cap, trace_pad;
  // wide mode: `wide` workgroups cooperate on one problem (small batches); per-slot barrier counters
  // (16 uints apart), reduction partials ([2][wide][12] doubles) and shared flags (8 ints)
  int32_t wide, wide_pad;
  uint32_t* wide_bar;
  double* wide_partials;
  int32_t* wide_flags;
  // sweep (mode 2)
  const double *sw_x, *sw_lam;
  double *sw_g, *sw_blocks;
};


}  // namespace dompc
