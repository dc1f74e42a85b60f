// dompc_device.hip - gfx950 entry points of the per-model code object.
// Built by do_mpc_amd/build.py:  hipcc --offload-arch=gfx950 -O3 --genco -include <model_gen.h> ...
// The same text is compiled by g++ -DDOMPC_HOST_EMU for the test-only host emulation (one "workgroup"
// = one host thread); see dompc_kernel.h.
#ifndef DOMPC_HOST_EMU
#include <hip/hip_runtime.h>
#define DOMPC_CONSTANT_TABLES 1          // structure tables of KArgs: constant address space (dompc_kargs.h)
#define DOMPC_FN __device__ static inline
#define DOMPC_CONST __device__ static constexpr
#define DOMPC_DEV __device__
#define DOMPC_HD __host__ __device__
#else
#include <stdlib.h>
#define DOMPC_FN static inline
#define DOMPC_CONST static const
#define DOMPC_DEV
#define DOMPC_HD
#endif

// resident wavefronts per SIMD the register allocation is sized for (2: 256 VGPRs, 3: 168, 4: 128)
#ifndef DOMPC_LB
#define DOMPC_LB 2
#endif
#ifndef DOMPC_SRC_DIGEST
#define DOMPC_SRC_DIGEST 0ULL
#endif

#include DOMPC_MODEL_HEADER
#include "dompc_kernel.h"

namespace dompc {
// sizes the generic runtime needs from the model-specific build
DOMPC_HD inline void model_info(const int32_t* in, int64_t* out) {
  // in: n_opt_x, n_g, n_edges, e_pad, n_nodes
  WsLayout L = ws_layout(in[0], in[1], in[2], in[3], in[4]);
  out[0] = NX; out[1] = NU; out[2] = NP; out[3] = NTVP; out[4] = NE; out[5] = NSE;
  out[6] = DEG; out[7] = NI; out[8] = M;
  out[9] = L.total;
  out[10] = SWEEP_BLOCK;
  out[11] = (int64_t)sizeof(KArgs);
  out[12] = RED_MAX; out[13] = ASM_N; out[14] = CUT1; out[15] = CUT2;      // exchange buffer layout (tree sharding)
  out[16] = DOMPC_SHARD;                                                   // built with tree-sharding support?
  out[17] = EL_SIZE;                                                       // LDS doubles per wavefront (edge / node working set)
  out[19] = QUAD_EDGE ? 4 : 1;                                             // edges per wavefront in the derivative sweep (4: dompc_quad.h)
  out[18] = (int64_t)DOMPC_SRC_DIGEST;                                     // digest of the kernel sources this object was compiled from (build.py)
}
}  // namespace dompc

#ifndef DOMPC_HOST_EMU
extern "C" __global__ void dompc_model_info_kernel(const int32_t* in, int64_t* out, char* hash) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    dompc::model_info(in, out);
    const char h[] = DOMPC_MODEL_HASH;
    for (int i = 0; i < (int)sizeof(h); ++i) hash[i] = h[i];
  }
}

#ifndef DOMPC_MAXBLOCK
#define DOMPC_MAXBLOCK 256         // largest workgroup the general code object is launched with (experiment: 512 = eight wavefronts per workgroup in wide mode)
#endif
extern "C" __global__ void __launch_bounds__(DOMPC_BLOCK_CONST ? DOMPC_BLOCK_CONST : DOMPC_MAXBLOCK, DOMPC_LB) dompc_solve_kernel(dompc::KArgs A) {
  using namespace dompc;
  const int POOL = A.pool_doubles;
  if (threadIdx.x < 32) lds_prof[threadIdx.x] = 0;
  if (threadIdx.x < 8) lds_flags[threadIdx.x] = 0;
  // defined LDS contents at kernel start (the pool of the previous kernel on this CU is still in there)
  for (int i = threadIdx.x; i < POOL; i += DOMPC_BDIM) lds_pool[i] = (i >= A.lds_fill_lo && i < A.lds_fill_hi) ? A.lds_fill : 0.0;
  for (int i = threadIdx.x; i < 2 * MAX_FILTER; i += DOMPC_BDIM) lds_filt[i] = 0.0;
  __syncthreads();
  // Thread context (make_thr).  Normal mode: one workgroup per problem, problems pulled from a device-wide counter.
  // Wide mode (small batches): K = A.wide workgroups per problem, static assignment problem = slot.
  const bool wide = dompc::WIDE_OK && (A.mode == 0 && A.wide > 1);
  const int slot = slot_of_block(A);
  if (wide && slot >= A.batch) return;
  Thr T = make_thr(A);
  T.kp = (const void*)__builtin_amdgcn_kernarg_segment_ptr();
  if (wide && A.mode != 1) dompc::xcd_census(T);
  if (A.mode == 1) {                     // Newton steps at one point for A.batch parameter vectors: one workgroup (= slot) each
    const int bq = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
    const int sq = slot_of_block(A);
    if (bq < A.batch) dompc::debug_newton(T, A, bq, sq);
    return;
  }
  // (single call site of solve_problem: the compiler inlines the whole solver into the kernel; an
  //  out-of-line copy keeps the context structs in scratch and is 2x slower)
  bool first = true;
  while (true) {
    int b;
    if (wide) {
      if (!first) break;
      b = slot;
    } else {
      if (threadIdx.x == 0) lds_b = atomicAdd(A.work_counter, 1);
      __syncthreads();
      b = lds_b;
      __syncthreads();
      if (b >= A.batch) break;
    }
    first = false;
    if (A.mode == 2) dompc::sweep_problem(T, A, b, slot);
    else dompc::solve_problem(T, A, b, slot);
  }
}
#else
extern "C" void dompc_hostemu_model_info(const int32_t* in, int64_t* out, char* hash) {
  dompc::model_info(in, out);
  const char h[] = DOMPC_MODEL_HASH;
  for (int i = 0; i < (int)sizeof(h); ++i) hash[i] = h[i];
}
extern "C" void dompc_hostemu_run(const dompc::KArgs* A) {
  static thread_local double red[dompc::RED_MAX];
  static thread_local double filt[2 * dompc::MAX_FILTER];
  static thread_local int flags[8];
  static thread_local double edge_lds[dompc::EL_SIZE];
  if (const char* poison = getenv("DOMPC_EMU_POISON")) {     // test aid: undefined "LDS" contents (read-before-write hunts)
    const double v = atof(poison);
    for (int i = 0; i < dompc::EL_SIZE; ++i) edge_lds[i] = v;
    for (int i = 0; i < dompc::RED_MAX; ++i) red[i] = v;
    for (int i = 0; i < 2 * dompc::MAX_FILTER; ++i) filt[i] = v;
  }
  dompc::Thr T{0, 1, red, filt, flags, edge_lds, nullptr, 1, 0, 1, 0, 1, nullptr, nullptr, 0u, 0u, dompc::make_xctx(*A), 0u, nullptr};
  if (A->mode == 1) {
    for (int b = 0; b < A->batch; ++b) dompc::debug_newton(T, *A, b, 0);
    return;
  }
  for (int b = 0; b < A->batch; ++b) {
    if (A->mode == 2) dompc::sweep_problem(T, *A, b, 0);
    else dompc::solve_problem(T, *A, b, 0);
  }
}
#endif
