// dompc_kargs.h - kernel argument block shared by the generic host runtime (dompc_runtime.cpp)
// and the per-model device code (dompc_kernel.h).  Plain data, no model-dependent sizes.
#pragma once
#include <stdint.h>
#include "../../include/dompc_ipm.h"

namespace dompc {
enum { WIDE_BAR_STRIDE = 64 };
enum { EP_PARENT = 0, EP_CHILD, EP_LEVEL, EP_WOFF, EP_PIDX, EP_ROW0, EP_XOFF_PARENT, EP_UOFF_PARENT, EP_XOFF_CHILD, EP_EPSOFF_PARENT,
       EP_OMEGA_LO, EP_OMEGA_HI, EP_N = 16 };
// Structure tables of the problem class: written once by dompc_create(), never by a kernel.  In device code they are
// pointers into the CONSTANT address space, which is what lets the compiler read them with scalar loads (s_load through
// the scalar cache, result in SGPRs) wherever the index is wave-uniform - with plain global pointers every look-up is a
// vector load and every address derived from it a 64-bit VGPR pair.  Same 64-bit pointers on the host side.
#if defined(DOMPC_CONSTANT_TABLES)      // (defined by dompc_device.hip; the host runtime sees ordinary pointers)
typedef const __attribute__((address_space(4))) int32_t* itab_t;
typedef const __attribute__((address_space(4))) double* dtab_t;
#else
typedef const int32_t* itab_t;
typedef const double* dtab_t;
#endif
struct KArgs {
  itab_t level_node_start, node_level, node_x_off, node_u_off, node_eps_off;
  itab_t node_child_start, node_child_count, node_parent, node_in_edge;
  itab_t edge_parent, edge_child, edge_pidx, edge_w_off, edge_row0, edge_level;
  dtab_t edge_omega;
  itab_t dummy_idx;
  // the indices an edge needs, resolved and side by side (EP_N ints per edge, written by dompc_create from the tables above): one
  // scalar load instead of seven look-ups of which three depend on the first - and one table pointer in registers instead of ten
  itab_t edge_pack;
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
  int32_t dbg_at_solution;   // mode 1: slacks of the nl_cons rows s = d(x), their multipliers mu / distance (a converged point
                             // of the barrier problem) instead of the pushed starting values
  const double *dbg_lam, *dbg_zl, *dbg_zu;
  double dbg_mu, dbg_delta;
  double *dbg_dx, *dbg_dlam, *dbg_rd, *dbg_c;
  // optional iteration trace of problem 0: 8 doubles per iteration (it, mu, E0, inf_pr, inf_du, alpha, delta_w, obj)
  double* trace;
  int32_t trace_cap, trace_pad;
  // wide mode: `wide` workgroups cooperate on one problem (small batches); per-slot barrier counters
  // (16 uints apart), reduction partials ([2][wide][12] doubles) and shared flags (8 ints)
  int32_t wide;
  int32_t pool_doubles;      // doubles in the dynamic LDS pool of a workgroup: max(waves * EL_SIZE, RED_MAX * threads)
  uint32_t* wide_bar;        // WIDE_BAR_STRIDE words per slot: [0] arrival counter, [1] XCD mask, [2] verdict, [8 + x] / [16 + x] / [24 + x]: arrival counter, release word and workgroup count of XCD x (two-level barrier)
  double* wide_partials;
  int32_t* wide_flags;
  // sweep (mode 2)
  const double *sw_x, *sw_lam;
  double *sw_g, *sw_blocks;
  // tree sharding of ONE problem over ranks; masks null = not sharded.
  // masks: 0 = another rank's, 1 = mine, 2 = replicated on every rank (counted once, by rank 0)
  const int8_t *x_mask, *g_mask, *e_mask, *n_mask;
  const int32_t* node_cut;   // index of a cut parent (a replicated node whose child edges are spread over the ranks), or -1
  int32_t n_cut, cut_level, shard_rank, shard_world;
  // exchange buffer (device memory, element-wise SUM over the ranks) and the handshake with the host service
  // loop: the kernel publishes a request (sequence number, element count) in pinned host memory and polls the
  // acknowledge word; the host runs the collective on the buffer in between.  Host emulation: direct callback.
  double* xbuf;
  int32_t xbuf_len;
  // wide mode, whole-chip placement: the K workgroups of problem slot s are the blocks s*K .. s*K + K - 1 - spread over ALL XCDs by the
  // dispatcher (block b on XCD b % 8) - instead of K blocks of one XCD; one large problem (a 243-leaf tree) then uses every CU of the chip.
  // The barrier notices the placement by itself (xcd_census) and keeps its L2 write-back.
  int32_t wide_spread;
  volatile uint32_t* x_req;
  volatile uint32_t* x_ack;
  volatile uint32_t* x_count;
  void (*x_callback)(void* ctx, double* buf, int32_t count);
  void* x_ctx;
  // stop request (watchdog / dompc_abort): a word in pinned host memory, read once per IPM iteration by thread 0 of the
  // problem and handed to the other threads through flags[6]; unfinished problems return status 6
  const int32_t* abort_flag;
  // debugging aid: value the LDS pool is filled with when the kernel starts (0 in production), words [lo, hi)
  double lds_fill;
  int32_t lds_fill_lo, lds_fill_hi;
  // bounds of opt_x as the solver uses them (relaxed; unused variables taken out): ONE copy for all problems of a launch - lbx / ubx
  // are inputs of the launch, not of a problem - instead of one per slot: the passes that read them (sweep, step rules, accept,
  // error measures) then hit the L2 instead of streaming 2 x n_opt_x doubles per slot from HBM.  Every problem writes the same
  // values at its start; null = per-slot copies (DOMPC_SHARED_BOUNDS=0)
  double *lb_sh, *ub_sh;
};


}  // namespace dompc
