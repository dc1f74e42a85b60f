// dompc_runtime.cpp - generic host runtime behind the C ABI of include/dompc_ipm.h.
//
// Owns: the loaded per-model gfx950 code object, device copies of the tree tables, the workspace
// slots of the persistent workgroups, staging buffers for the host-pointer entry points.
// It contains no model-dependent code: sizes come from the code object (dompc_model_info_kernel).
//
// Build flavours (do_mpc_amd/build.py):
//   product : hipcc -shared -fPIC dompc_runtime.cpp            -> libdompc_ipm.so   (HIP only)
//   test    : g++ -DDOMPC_HOST_EMU dompc_runtime.cpp dompc_device.hip(as C++) -> tests/_hostemu/*.so
//             ("device" memory = host memory, a workgroup = the calling thread).  Never shipped.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <cmath>
#include <string>
#include <thread>
#include <vector>

#include "dompc_kargs.h"

#ifndef DOMPC_HOST_EMU
#include <hip/hip_runtime.h>
#else
extern "C" void dompc_hostemu_model_info(const int32_t* in, int64_t* out, char* hash);
extern "C" void dompc_hostemu_run(const dompc::KArgs* A);
#endif

static thread_local std::string g_create_error;
static const int BATCH_ONE_WAVE = 4096;     // batch size from which a problem gets one wavefront instead of four (see dompc_create)

struct dompc_handle {
  dompc_problem_desc d;
  std::string error;
  std::string code_path;
  int32_t e_pad = 0, n_slots = 0, block = 256, occupancy = 0, n_leaves = 1;
  bool batch_object_stale = false;       // a `_batch` sibling exists but was built from other sources / another model: not used
  bool block_auto = true;          // threads per problem chosen per call from the batch size
  int32_t slots64 = 0, slots256 = 0;   // resident workgroups at 64 / 256 threads
  int64_t ws_stride = 0, sweep_block = 0, el_size = 0, edges_per_wave = 1;
  dompc::KArgs base;       // tables + workspace filled in, I/O pointers zero
  std::vector<void*> dev_allocs;
  // staging for host-pointer calls
  double *s_x0 = nullptr, *s_lbx = nullptr, *s_ubx = nullptr, *s_lbg = nullptr, *s_ubg = nullptr, *s_p = nullptr;
  double *s_x = nullptr, *s_g = nullptr, *s_lamx = nullptr, *s_lamg = nullptr, *s_f = nullptr;
  dompc_stats* s_stats = nullptr;
  double *s_dbg[8] = {nullptr};
  double* s_trace = nullptr;
  int32_t trace_cap = 4096;
  int32_t cap_batch = 0;
  // tree sharding
  int64_t xlayout[4] = {0, 0, 0, 0};   // RED_MAX, ASM_N, CUT1, CUT2 of the code object
  bool sharded = false, shard_capable = false;
  dompc_allreduce_fn allreduce = nullptr;
  void* allreduce_ctx = nullptr;
  uint32_t* x_words = nullptr;           // pinned host memory: [req, ack, count, off] (device build)
  int32_t* abort_word = nullptr;         // pinned host memory (device build) / plain word: stop request read by the kernel
  double watchdog_s = 600.0;
  int64_t n_exchanges = 0;                      // cross-rank exchanges served during the last sharded solve
#ifndef DOMPC_HOST_EMU
  // native RCCL collective (dlopen'ed): communicator of the sharded problem and its stream
  struct RcclUid { char internal[128]; };
  void* rccl_lib = nullptr;
  int (*nccl_get_unique_id)(RcclUid*) = nullptr;
  int (*nccl_comm_init_rank)(void**, int, RcclUid, int) = nullptr;
  int (*nccl_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*nccl_comm_destroy)(void*) = nullptr;
  const char* (*nccl_error_string)(int) = nullptr;
  void* rccl_comm = nullptr;
  hipStream_t rccl_stream = nullptr;
#endif
#ifndef DOMPC_HOST_EMU
  hipModule_t module = nullptr, module_batch = nullptr;
  hipFunction_t fn_solve = nullptr, fn_info = nullptr, fn_solve_batch = nullptr;
  hipStream_t stream = nullptr;
  hipStream_t shard_stream = nullptr;    // lowest priority: never shares a hardware queue with the collective's kernels
#endif
};

// ------------------------------------------------------------------------------------------------
#ifndef DOMPC_HOST_EMU
#define HIPCHK(h, expr)                                                                     \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      (h)->error = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
      return 1;                                                                             \
    }                                                                                       \
  } while (0)

static int dev_alloc(dompc_handle* h, void** p, size_t bytes) {
  if (bytes == 0) bytes = 8;
  HIPCHK(h, hipMalloc(p, bytes));
  h->dev_allocs.push_back(*p);
  return 0;
}
static void dev_release(dompc_handle* h, void* p) {
  if (!p) return;
  for (size_t i = 0; i < h->dev_allocs.size(); ++i)
    if (h->dev_allocs[i] == p) { h->dev_allocs.erase(h->dev_allocs.begin() + i); hipFree(p); return; }
}
static int h2d(dompc_handle* h, void* dst, const void* src, size_t bytes) {
  if (!bytes) return 0;
  HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
  return 0;
}
static int d2h(dompc_handle* h, void* dst, const void* src, size_t bytes) {
  if (!bytes) return 0;
  HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
  return 0;
}
static int dev_sync(dompc_handle* h) {
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
// Wait for `st` with the watchdog: after watchdog_s the stop request is raised (the kernel leaves its IPM loops with
// status 6); if the stream still has not drained after a grace period the call fails instead of blocking forever.
// The limit is per ROUND over the resident problem slots (a batch of B problems on n_slots slots takes ceil(B / n_slots) rounds): a
// legitimately long batch is not cut short by a limit that was sized for one problem.  A watchdog that fired leaves a message in
// dompc_last_error even though the call succeeds (the problems it stopped carry status 6, `User_Requested_Stop`).
static int dev_sync_watchdog(dompc_handle* h, hipStream_t st, int rounds = 1) {
  const auto t0 = std::chrono::steady_clock::now();
  bool raised = false;
  const double limit = h->watchdog_s * (rounds > 1 ? rounds : 1);
  while (true) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) { h->error = std::string("hipStreamQuery: ") + hipGetErrorString(q); return 1; }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!raised && dt > limit) { int32_t z_ = 0; __atomic_compare_exchange_n(h->abort_word, &z_, 1, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE); raised = true; }   // (0 -> 1: a user's request, 2, is left alone)
    if (raised && dt > limit + 30.0) {
      h->error = "watchdog: the solver kernel did not finish and did not react to the stop request";
      return 1;
    }
    if (dt > 0.002) std::this_thread::sleep_for(std::chrono::microseconds(dt > 0.2 ? 500 : 20));
  }
  if (raised) {
    // only the watchdog's own request is withdrawn (compare-exchange 1 -> 0: a user's dompc_abort(h, 1) writes 2 and stays -
    // it is documented as sticky until dompc_abort(h, 0)); the caller learns about it through the error string and status 6
    int32_t expect = 1;
    __atomic_compare_exchange_n(h->abort_word, &expect, 0, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
    char msg[200];
    snprintf(msg, sizeof msg, "watchdog: stop request raised after %.0f s (DOMPC_WATCHDOG_S = %.0f s per round over the problem slots, %d round(s)); "
             "unfinished problems return status 6", limit, h->watchdog_s, rounds > 1 ? rounds : 1);
    h->error = msg;
  }
  return 0;
}
static int dev_zero(dompc_handle* h, void* p, size_t bytes, hipStream_t s) {
  HIPCHK(h, hipMemsetAsync(p, 0, bytes, s));
  return 0;
}
#else
static int dev_alloc(dompc_handle* h, void** p, size_t bytes) {
  if (bytes == 0) bytes = 8;
  *p = calloc(1, bytes);
  if (!*p) { h->error = "out of memory"; return 1; }
  h->dev_allocs.push_back(*p);
  return 0;
}
static void dev_release(dompc_handle* h, void* p) {
  if (!p) return;
  for (size_t i = 0; i < h->dev_allocs.size(); ++i)
    if (h->dev_allocs[i] == p) { h->dev_allocs.erase(h->dev_allocs.begin() + i); free(p); return; }
}
static int h2d(dompc_handle*, void* dst, const void* src, size_t bytes) { if (bytes) memcpy(dst, src, bytes); return 0; }
static int d2h(dompc_handle*, void* dst, const void* src, size_t bytes) { if (bytes) memcpy(dst, src, bytes); return 0; }
static int dev_sync(dompc_handle*) { return 0; }
#endif

extern "C" int dompc_abort(dompc_handle* h, int32_t stop) {
  if (!h || !h->abort_word) return 1;
  __atomic_store_n(h->abort_word, stop ? 2 : 0, __ATOMIC_RELEASE);     // (2: the user's request - the watchdog only withdraws its own 1)
  return 0;
}

template <typename Tp>
static int upload(dompc_handle* h, const Tp** dst, const Tp* src, size_t n) {
  void* p = nullptr;
  if (dev_alloc(h, &p, n * sizeof(Tp))) return 1;
  if (n && !src) { h->error = "null table pointer in problem description"; return 1; }
  if (h2d(h, p, src, n * sizeof(Tp))) return 1;
  *dst = (const Tp*)p;
  return 0;
}

// ------------------------------------------------------------------------------------------------
extern "C" void dompc_default_options(dompc_options* o) {
  o->tol = 1e-8; o->dual_inf_tol = 1.0; o->constr_viol_tol = 1e-4; o->compl_inf_tol = 1e-4;
  o->acceptable_tol = 1e-6; o->mu_init = 0.1; o->kappa_mu = 0.2; o->theta_mu = 1.5; o->kappa_eps = 10.0;
  o->tau_min = 0.99; o->bound_push = 0.01; o->bound_frac = 0.01; o->bound_relax_factor = 1e-8;
  o->nlp_scaling_max_gradient = 100.0; o->delta_w_0 = 1e-4; o->delta_w_min = 1e-20; o->delta_w_max = 1e20;
  o->kappa_w_minus = 1.0 / 3.0; o->kappa_w_plus = 8.0; o->kappa_w_plus_bar = 100.0;
  o->max_iter = 3000; o->acceptable_iter = 15; o->obj_scaling = 1; o->max_soc = 4;
  o->constr_mult_init_max = 1000.0;
  o->watchdog_shortened_iter_trigger = 10; o->watchdog_trial_iter_max = 3;
}

extern "C" const char* dompc_status_string(int32_t s) {
  switch (s) {
    case 0: return "Solve_Succeeded";
    case 1: return "Solved_To_Acceptable_Level";
    case 2: return "Maximum_Iterations_Exceeded";
    case 3: return "Error_In_Step_Computation";
    case 4: return "Invalid_Number_Detected";
    case 6: return "User_Requested_Stop";
    default: return "Internal_Error";
  }
}

extern "C" const char* dompc_last_error(const dompc_handle* h) { return h ? h->error.c_str() : g_create_error.c_str(); }

extern "C" void dompc_destroy(dompc_handle* h) {
  if (!h) return;
#ifndef DOMPC_HOST_EMU
  hipSetDevice(h->d.device);
  for (void* p : h->dev_allocs) hipFree(p);
  if (h->module) hipModuleUnload(h->module);
  if (h->module_batch) hipModuleUnload(h->module_batch);
  if (h->stream) hipStreamDestroy(h->stream);
  if (h->x_words) hipHostFree(h->x_words);
  if (h->abort_word) hipHostFree(h->abort_word);
  if (h->shard_stream) hipStreamDestroy(h->shard_stream);
  if (h->rccl_comm && h->nccl_comm_destroy) h->nccl_comm_destroy(h->rccl_comm);
  if (h->rccl_stream) hipStreamDestroy(h->rccl_stream);
#else
  for (void* p : h->dev_allocs) free(p);
  free(h->abort_word);
#endif
  delete h;
}

static int ensure_staging(dompc_handle* h, int B) {
  if (B <= h->cap_batch) return 0;
  const dompc_problem_desc& d = h->d;
  // (re)allocate for the larger batch; the superseded per-batch buffers are released now
  if (dev_sync(h)) return 1;
  for (void* old : {(void*)h->s_x0, (void*)h->s_p, (void*)h->s_x, (void*)h->s_g, (void*)h->s_lamx, (void*)h->s_lamg,
                    (void*)h->s_f, (void*)h->s_stats})
    dev_release(h, old);
  if (dev_alloc(h, (void**)&h->s_x0, sizeof(double) * (size_t)B * d.n_opt_x)) return 1;
  if (dev_alloc(h, (void**)&h->s_p, sizeof(double) * (size_t)B * d.n_opt_p)) return 1;
  if (dev_alloc(h, (void**)&h->s_x, sizeof(double) * (size_t)B * d.n_opt_x)) return 1;
  if (dev_alloc(h, (void**)&h->s_g, sizeof(double) * (size_t)B * d.n_g)) return 1;
  if (dev_alloc(h, (void**)&h->s_lamx, sizeof(double) * (size_t)B * d.n_opt_x)) return 1;
  if (dev_alloc(h, (void**)&h->s_lamg, sizeof(double) * (size_t)B * d.n_g)) return 1;
  if (dev_alloc(h, (void**)&h->s_f, sizeof(double) * (size_t)B)) return 1;
  if (dev_alloc(h, (void**)&h->s_stats, sizeof(dompc_stats) * (size_t)B)) return 1;
  if (!h->s_lbx) {
    if (dev_alloc(h, (void**)&h->s_lbx, sizeof(double) * d.n_opt_x)) return 1;
    if (dev_alloc(h, (void**)&h->s_ubx, sizeof(double) * d.n_opt_x)) return 1;
    if (dev_alloc(h, (void**)&h->s_lbg, sizeof(double) * d.n_g)) return 1;
    if (dev_alloc(h, (void**)&h->s_ubg, sizeof(double) * d.n_g)) return 1;
  }
  h->cap_batch = B;
  return 0;
}

static void* main_stream(dompc_handle* h) {      // the stream of the staging copies
#ifndef DOMPC_HOST_EMU
  return (void*)h->stream;
#else
  (void)h;
  return nullptr;
#endif
}
static void* own_stream(dompc_handle* h) {
#ifndef DOMPC_HOST_EMU
  return (void*)((h->sharded && h->shard_stream) ? h->shard_stream : h->stream);
#else
  (void)h;
  return nullptr;
#endif
}

// largest workgroup size <= `block` whose LDS pool (one edge working set per wavefront) fits the 160 KiB of a CU - models with
// a large dense edge working set (DAE path: 45 kB per wavefront for the double inverted pendulum) run with fewer wavefronts
// per workgroup
static int fit_block(const dompc_handle* h, int block) {
  const int64_t lds_max = 160 * 1024 - 4096;                     // (static LDS of the kernel: filter, flags, counters)
  while (block > 64 && (int64_t)(block / 64) * h->el_size * (int64_t)sizeof(double) > lds_max) block /= 2;
  return block;
}

// `block` threads per workgroup (a multiple of 64): the LDS pool is sized for block/64 wavefronts
static int launch(dompc_handle* h, dompc::KArgs& A, int grid, int block, void* stream_v) {
#ifndef DOMPC_HOST_EMU
  hipStream_t st = (hipStream_t)stream_v;          // nullptr = HIP default stream
  if (dev_zero(h, A.work_counter, sizeof(int32_t), st)) return 1;
  const int64_t per_wave = (int64_t)(block / 64) * h->el_size, red = h->xlayout[0] * (int64_t)block;
  A.pool_doubles = (int32_t)(per_wave > red ? per_wave : red);
  size_t sz = sizeof(A);
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  // (batch launches of the solver with one 64-thread workgroup per problem, not sharded, run the build of the kernels that is compiled for
  //  exactly that shape when it was loaded: build.py batch_only)
  hipFunction_t fn = (h->fn_solve_batch && (A.mode == 0 || A.mode == 2) && A.wide <= 1 && block == 64 && !h->sharded) ? h->fn_solve_batch : h->fn_solve;
  HIPCHK(h, hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, (unsigned)(A.pool_doubles * sizeof(double)), st, nullptr, cfg));
#else
  (void)grid; (void)block; (void)stream_v;
  dompc_hostemu_run(&A);
#endif
  return 0;
}

extern "C" int dompc_create(const dompc_problem_desc* desc, dompc_handle** out) {
  if (!desc || !out) { g_create_error = "null argument"; return 1; }
  dompc_handle* h = new dompc_handle();
  h->d = *desc;
  auto fail = [&](int) { g_create_error = h->error; dompc_destroy(h); *out = nullptr; return 1; };
  const dompc_problem_desc& d = h->d;
  if (d.n_edges <= 0 || d.n_nodes <= 0 || d.n_opt_x <= 0) { h->error = "empty problem description"; return fail(1); }
  // threads per problem in batch mode: one wavefront per problem (4x the problem slots, no workgroup barriers, no idle
  // wavefronts in the tree recursion) once the batch fills every resident wavefront at least twice, else four
  // wavefronts per problem (measured on MI355X, industrial_poly: B = 1024: 256 threads 1832 vs 64 threads 1152
  // steps/s; B = 4096: 3093 vs 3221)
  // The choice is made per call from the batch size unless the description fixes it (block_auto).
  h->block = d.block_threads > 0 ? d.block_threads : 256;
  h->block_auto = d.block_threads <= 0;
  if (const char* be = getenv("DOMPC_BLOCK")) { h->block = atoi(be); h->block_auto = false; }   // tuning aid (64/128/256)
  if (h->block != 64 && h->block != 128 && h->block != 256) { h->error = "block_threads must be 64, 128 or 256"; return fail(1); }
#ifndef DOMPC_HOST_EMU
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    h->error = "no HIP device available: the dompc IPM backend requires an AMD GPU (gfx950)";
    return fail(1);
  }
  if (hipSetDevice(d.device) != hipSuccess) { h->error = "hipSetDevice failed"; return fail(1); }
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { h->error = "hipStreamCreate failed"; return fail(1); }
  if (hipHostMalloc((void**)&h->abort_word, 64, hipHostMallocMapped) != hipSuccess) { h->error = "hipHostMalloc failed"; return fail(1); }
  *h->abort_word = 0;
  if (!d.code_object_path) { h->error = "code_object_path is null"; return fail(1); }
  h->code_path = d.code_object_path;
  if (hipModuleLoad(&h->module, h->code_path.c_str()) != hipSuccess) {
    h->error = "hipModuleLoad failed for " + h->code_path;
    return fail(1);
  }
  {
    // optional sibling `<name>_batch.hsaco` (build.py: batch_only): the same kernels compiled for "one 64-thread workgroup per problem" only
    std::string bp = h->code_path;
    const size_t dot = bp.rfind(".hsaco");
    if (dot != std::string::npos && !getenv("DOMPC_NO_BATCH_OBJECT")) {
      bp.insert(dot, "_batch");
      FILE* f = fopen(bp.c_str(), "rb");
      if (f) {
        fclose(f);
        if (hipModuleLoad(&h->module_batch, bp.c_str()) != hipSuccess ||
            hipModuleGetFunction(&h->fn_solve_batch, h->module_batch, "dompc_solve_kernel") != hipSuccess) {
          h->error = "hipModuleLoad failed for " + bp;
          return fail(1);
        }
      }
    }
  }
  if (hipModuleGetFunction(&h->fn_solve, h->module, "dompc_solve_kernel") != hipSuccess ||
      hipModuleGetFunction(&h->fn_info, h->module, "dompc_model_info_kernel") != hipSuccess) {
    h->error = "code object lacks dompc kernels: " + h->code_path;
    return fail(1);
  }
#else
  h->block = 1;
  h->abort_word = (int32_t*)calloc(16, sizeof(int32_t));
#endif
  if (const char* wd = getenv("DOMPC_WATCHDOG_S")) h->watchdog_s = atof(wd);
  h->e_pad = ((d.n_edges + 15) / 16) * 16;
  // ---- model info from the code object
  int32_t in_h[5] = {d.n_opt_x, d.n_g, d.n_edges, h->e_pad, d.n_nodes};
  int64_t info[20] = {0};
  char hash[64] = {0};
#ifndef DOMPC_HOST_EMU
  auto query_info = [&](hipFunction_t fn, int64_t* info_o, char* hash_o) -> int {
    int32_t* in_d; int64_t* out_d; char* hash_d;
    if (dev_alloc(h, (void**)&in_d, sizeof(in_h)) || dev_alloc(h, (void**)&out_d, sizeof(info)) ||
        dev_alloc(h, (void**)&hash_d, sizeof(hash))) return 1;
    if (h2d(h, in_d, in_h, sizeof(in_h))) return 1;
    if (dev_zero(h, out_d, sizeof(info), h->stream)) return 1;
    struct { const int32_t* a; int64_t* b; char* c; } args = {in_d, out_d, hash_d};
    size_t sz = sizeof(args);
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    if (hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, h->stream, nullptr, cfg) != hipSuccess) {
      h->error = "launch of dompc_model_info_kernel failed"; return 1;
    }
    if (d2h(h, info_o, out_d, sizeof(info)) || d2h(h, hash_o, hash_d, sizeof(hash)) || dev_sync(h)) return 1;
    return 0;
  };
  if (query_info(h->fn_info, info, hash)) return fail(1);
  if (h->fn_solve_batch) {
    // The sibling is only ever launched with the general object's argument block and workspace layout: it must come from the same
    // sources and the same model (a handle with a small max_batch rebuilds only the general object after a kernel change;
    // a stale sibling with another KArgs layout would then be launched for every 64-thread batch).  A sibling that does not match is
    // not used - the general object runs those launches instead.
    int64_t info_b[20] = {0};
    char hash_b[64] = {0};
    hipFunction_t fn_info_b = nullptr;
    bool same = hipModuleGetFunction(&fn_info_b, h->module_batch, "dompc_model_info_kernel") == hipSuccess && !query_info(fn_info_b, info_b, hash_b);
    for (int i = 0; same && i < 20; ++i) same = info_b[i] == info[i];
    same = same && strncmp(hash, hash_b, 63) == 0;
    if (!same) {
      hipModuleUnload(h->module_batch);
      h->module_batch = nullptr;
      h->fn_solve_batch = nullptr;
      h->batch_object_stale = true;
      h->error.clear();
    }
  }
#else
  dompc_hostemu_model_info(in_h, info, hash);
#endif
  const int64_t want[9] = {d.nx, d.nu, d.np, d.ntvp, d.ne, d.ns, d.M == 0 ? 0 : d.deg, d.M == 0 ? 1 : d.ni, d.M};
  for (int i = 0; i < 9; ++i)
    if (info[i] != want[i]) {
      char buf[256];
      snprintf(buf, sizeof(buf), "code object was built for different model dimensions (field %d: %lld vs %lld)", i,
               (long long)info[i], (long long)want[i]);
      h->error = buf;
      return fail(1);
    }
  if (info[11] != (int64_t)sizeof(dompc::KArgs)) { h->error = "KArgs layout mismatch between runtime and code object"; return fail(1); }
  if (d.model_hash && strncmp(d.model_hash, hash, 63) != 0) {
    h->error = std::string("model hash mismatch: code object ") + hash + " vs description " + d.model_hash;
    return fail(1);
  }
  h->ws_stride = info[9];
  h->sweep_block = info[10];
  for (int i = 0; i < 4; ++i) h->xlayout[i] = info[12 + i];
  h->shard_capable = info[16] != 0;
  h->el_size = info[17];
  h->edges_per_wave = info[19] > 0 ? info[19] : 1;
  // ---- slots: one per workgroup the device can keep resident (occupancy of the solver kernel at this block size
  //      and LDS pool, times the number of CUs); more problems than slots are pulled from a work counter
  int max_batch = d.max_batch > 0 ? d.max_batch : 1;
  auto resident_at = [&](int block) -> int {
    int resident = 512 * (256 / block);
#ifndef DOMPC_HOST_EMU
    const int64_t per_wave = (int64_t)(block / 64) * h->el_size, red = h->xlayout[0] * (int64_t)block;
    const size_t lds = sizeof(double) * (size_t)(per_wave > red ? per_wave : red);
    int occ = 0, cus = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ, h->fn_solve, block, lds) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d.device) == hipSuccess && occ > 0 && cus > 0)
      resident = occ * cus;
    h->occupancy = occ;
#endif
    return resident;
  };
  h->slots64 = resident_at(64);
  h->slots256 = resident_at(fit_block(h, 256));
  {
    // batches of >= BATCH_ONE_WAVE problems run one wavefront per problem and need that many more slots
    const int resident = h->block_auto ? (max_batch >= BATCH_ONE_WAVE ? h->slots64 : h->slots256) : resident_at(h->block);
    h->n_slots = d.n_slots > 0 ? d.n_slots : (max_batch < resident ? max_batch : resident);
  }
  if (const char* se = getenv("DOMPC_SLOTS")) { const int v = atoi(se); if (v >= 1) h->n_slots = v; }      // (tuning aid; nonsense values are ignored)
  if (h->n_slots < 1) h->n_slots = 1;
#ifdef DOMPC_HOST_EMU
  h->n_slots = 1;
#endif
  // ---- tables
  dompc::KArgs& A = h->base;
  memset(&A, 0, sizeof(A));
  int rc = 0;
  rc |= upload(h, &A.level_node_start, d.level_node_start, d.N + 2);
  rc |= upload(h, &A.node_level, d.node_level, d.n_nodes);
  rc |= upload(h, &A.node_x_off, d.node_x_off, d.n_nodes);
  rc |= upload(h, &A.node_u_off, d.node_u_off, d.n_nodes);
  rc |= upload(h, &A.node_eps_off, d.node_eps_off, d.n_nodes);
  rc |= upload(h, &A.node_child_start, d.node_child_start, d.n_nodes);
  rc |= upload(h, &A.node_child_count, d.node_child_count, d.n_nodes);
  rc |= upload(h, &A.node_parent, d.node_parent, d.n_nodes);
  rc |= upload(h, &A.node_in_edge, d.node_in_edge, d.n_nodes);
  rc |= upload(h, &A.edge_parent, d.edge_parent, d.n_edges);
  rc |= upload(h, &A.edge_child, d.edge_child, d.n_edges);
  rc |= upload(h, &A.edge_pidx, d.edge_pidx, d.n_edges);
  rc |= upload(h, &A.edge_w_off, d.edge_w_off, d.n_edges);
  rc |= upload(h, &A.edge_row0, d.edge_row0, d.n_edges);
  rc |= upload(h, &A.edge_level, d.edge_level, d.n_edges);
  rc |= upload(h, &A.edge_omega, d.edge_omega, d.n_edges);
  rc |= upload(h, &A.dummy_idx, d.dummy_idx, d.n_dummy);
  if (rc) return fail(1);
  {
    std::vector<int32_t> pack((size_t)d.n_edges * dompc::EP_N, 0);
    for (int e = 0; e < d.n_edges; ++e) {
      int32_t* q = pack.data() + (size_t)e * dompc::EP_N;
      const int n = d.edge_parent[e], cn = d.edge_child[e];
      q[dompc::EP_PARENT] = n; q[dompc::EP_CHILD] = cn; q[dompc::EP_LEVEL] = d.edge_level[e]; q[dompc::EP_WOFF] = d.edge_w_off[e];
      q[dompc::EP_PIDX] = d.edge_pidx[e]; q[dompc::EP_ROW0] = d.edge_row0[e];
      q[dompc::EP_XOFF_PARENT] = d.node_x_off[n]; q[dompc::EP_UOFF_PARENT] = d.node_u_off[n]; q[dompc::EP_XOFF_CHILD] = d.node_x_off[cn];
      q[dompc::EP_EPSOFF_PARENT] = d.node_eps_off ? d.node_eps_off[n] : -1;
      int64_t ob;
      memcpy(&ob, &d.edge_omega[e], sizeof ob);
      q[dompc::EP_OMEGA_LO] = (int32_t)(uint32_t)(ob & 0xffffffffll); q[dompc::EP_OMEGA_HI] = (int32_t)(uint32_t)((uint64_t)ob >> 32);
    }
    if (upload(h, &A.edge_pack, pack.data(), pack.size())) return fail(1);
  }
  A.N = d.N; A.n_nodes = d.n_nodes; A.n_edges = d.n_edges; A.n_dummy = d.n_dummy;
  A.n_opt_x = d.n_opt_x; A.n_opt_p = d.n_opt_p; A.n_g = d.n_g; A.e_pad = h->e_pad;
  A.p_off_tvp = d.p_off_tvp; A.p_off_p = d.p_off_p; A.p_off_uprev = d.p_off_uprev;
  {
    // chain_level: first stage from which every node (k, s) has exactly one child, (k+1, s), over edge node_child_start[(k, 0)] + s
    // (the chain walks of the Riccati passes form these indices arithmetically)
    const int32_t* ls = desc->level_node_start;
    const int S = ls[d.N + 1] - ls[d.N];
    h->n_leaves = S > 0 ? S : 1;
    int cl = d.N;
    for (int k = d.N - 1; k >= 0; --k) {
      bool ok = (ls[k + 1] - ls[k]) == S;
      for (int s = 0; ok && s < S; ++s) {
        const int n = ls[k] + s;
        ok = desc->node_child_count[n] == 1 && desc->edge_child[desc->node_child_start[n]] == ls[k + 1] + s &&
             desc->node_child_start[n] == desc->node_child_start[ls[k]] + s;
      }
      if (!ok) break;
      cl = k;
    }
    A.chain_level = cl;
  }
  A.opt = d.opts;
  if (const char* mi = getenv("DOMPC_MAX_ITER")) A.opt.max_iter = atoi(mi);      // debugging aid
  if (const char* lf = getenv("DOMPC_LDS_FILL")) {                                 // debugging aid: "value[,lo,hi]"
    double v = 0.0; int lo = 0, hi = 1 << 30;
    sscanf(lf, "%lf,%d,%d", &v, &lo, &hi);
    A.lds_fill = v; A.lds_fill_lo = lo; A.lds_fill_hi = hi;
  }
  A.n_slots = h->n_slots;
  A.ws_stride = h->ws_stride;
#ifndef DOMPC_HOST_EMU
  {
    void* dw = nullptr;
    if (hipHostGetDevicePointer(&dw, h->abort_word, 0) != hipSuccess) { h->error = "hipHostGetDevicePointer failed"; return fail(1); }
    A.abort_flag = (const int32_t*)dw;
  }
#else
  A.abort_flag = h->abort_word;
#endif
  if (dev_alloc(h, (void**)&A.ws, sizeof(double) * (size_t)h->ws_stride * h->n_slots)) return fail(1);
#ifndef DOMPC_HOST_EMU
  if (const char* fill = getenv("DOMPC_WS_FILL")) {      // debugging aid: poison the workspace (uninitialised reads)
    if (hipMemset(A.ws, atoi(fill), sizeof(double) * (size_t)h->ws_stride * h->n_slots) != hipSuccess) { h->error = "hipMemset failed"; return fail(1); }
  }
#endif
  if (dev_alloc(h, (void**)&A.work_counter, 64)) return fail(1);
  A.lb_sh = A.ub_sh = nullptr;
  {
    const char* sb = getenv("DOMPC_SHARED_BOUNDS");      // (measurement aid: 0 = per-slot copies of the bounds)
    if (!sb || atoi(sb) != 0) {
      if (dev_alloc(h, (void**)&A.lb_sh, sizeof(double) * (size_t)d.n_opt_x)) return fail(1);
      if (dev_alloc(h, (void**)&A.ub_sh, sizeof(double) * (size_t)d.n_opt_x)) return fail(1);
    }
  }
  // wide mode (small batches): up to 64 slots x 32 workgroups
  if (dev_alloc(h, (void**)&A.wide_bar, sizeof(uint32_t) * dompc::WIDE_BAR_STRIDE * 64)) return fail(1);
  if (dev_alloc(h, (void**)&A.wide_flags, sizeof(int32_t) * 8 * 64)) return fail(1);
  if (dev_alloc(h, (void**)&A.wide_partials, sizeof(double) * 64 * 2 * 32 * 12)) return fail(1);
  for (int i = 0; i < 8; ++i)
    if (dev_alloc(h, (void**)&h->s_dbg[i], sizeof(double) * (size_t)(d.n_opt_x > d.n_g ? d.n_opt_x : d.n_g))) return fail(1);
  if (dev_alloc(h, (void**)&h->s_trace, sizeof(double) * 8 * h->trace_cap)) return fail(1);
  A.trace = h->s_trace; A.trace_cap = h->trace_cap;
  if (const char* xt = getenv("DOMPC_EXTRA_TRAFFIC")) A.trace_pad = atoi(xt);    // measurement aid, see sweep()
  if (dev_sync(h)) return fail(1);
  // the description's table pointers are not valid after return
  h->d.level_node_start = nullptr;
  *out = h;
  return 0;
}

extern "C" int64_t dompc_workspace_bytes(const dompc_handle* h) { return h ? (int64_t)sizeof(double) * h->ws_stride * h->n_slots : 0; }
extern "C" int32_t dompc_num_slots(const dompc_handle* h) { return h ? h->n_slots : 0; }
extern "C" int64_t dompc_sweep_block_doubles(const dompc_handle* h) { return h ? h->sweep_block : 0; }

extern "C" int64_t dompc_exchange_doubles(const dompc_handle* h, int32_t world, int32_t n_cut) {
  if (!h || world < 1 || n_cut < 0) return 0;
  const int64_t* L = h->xlayout;
  return (int64_t)world * L[0] + (int64_t)n_cut * (L[1] + L[2] + L[3]) + 2 * (int64_t)world;
}

#ifndef DOMPC_HOST_EMU
static int rccl_open(dompc_handle* h, const char* path) {
  if (h->rccl_lib) return 0;
  h->rccl_lib = dlopen(path && *path ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h->rccl_lib) { h->error = std::string("dlopen(librccl) failed: ") + dlerror(); return 1; }
  h->nccl_get_unique_id = (int (*)(dompc_handle::RcclUid*))dlsym(h->rccl_lib, "ncclGetUniqueId");
  h->nccl_comm_init_rank = (int (*)(void**, int, dompc_handle::RcclUid, int))dlsym(h->rccl_lib, "ncclCommInitRank");
  h->nccl_all_reduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h->rccl_lib, "ncclAllReduce");
  h->nccl_comm_destroy = (int (*)(void*))dlsym(h->rccl_lib, "ncclCommDestroy");
  h->nccl_error_string = (const char* (*)(int))dlsym(h->rccl_lib, "ncclGetErrorString");
  if (!h->nccl_get_unique_id || !h->nccl_comm_init_rank || !h->nccl_all_reduce || !h->nccl_comm_destroy) {
    h->error = "librccl does not export the expected nccl* entry points"; return 1;
  }
  return 0;
}
#define NCCLCHK(h, expr)                                                                                       \
  do {                                                                                                         \
    const int _r = (expr);                                                                                     \
    if (_r != 0) {                                                                                             \
      (h)->error = std::string(#expr) + ": " + ((h)->nccl_error_string ? (h)->nccl_error_string(_r) : "rccl error"); \
      return 1;                                                                                                \
    }                                                                                                          \
  } while (0)
#endif

extern "C" int dompc_rccl_unique_id(dompc_handle* h, const char* librccl_path, uint8_t id[128]) {
  if (!h || !id) return 1;
#ifndef DOMPC_HOST_EMU
  if (rccl_open(h, librccl_path)) return 1;
  dompc_handle::RcclUid u;
  NCCLCHK(h, h->nccl_get_unique_id(&u));
  memcpy(id, u.internal, 128);
  return 0;
#else
  (void)librccl_path; h->error = "no RCCL in the host emulation"; return 1;
#endif
}

extern "C" int dompc_rccl_init(dompc_handle* h, const char* librccl_path, const uint8_t id[128], int32_t rank, int32_t world) {
  if (!h || !id) return 1;
#ifndef DOMPC_HOST_EMU
  if (rccl_open(h, librccl_path)) return 1;
  HIPCHK(h, hipSetDevice(h->d.device));
  if (h->rccl_comm) { h->nccl_comm_destroy(h->rccl_comm); h->rccl_comm = nullptr; }
  dompc_handle::RcclUid u;
  memcpy(u.internal, id, 128);
  NCCLCHK(h, h->nccl_comm_init_rank(&h->rccl_comm, world, u, rank));
  if (!h->rccl_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->rccl_stream, hipStreamNonBlocking));
  return 0;
#else
  (void)librccl_path; (void)rank; (void)world; h->error = "no RCCL in the host emulation"; return 1;
#endif
}

extern "C" int dompc_set_sharding(dompc_handle* h, const dompc_shard_desc* s) {
  if (!h) return 1;
  dompc::KArgs& A = h->base;
  if (!s) {
    A.x_mask = A.g_mask = A.e_mask = A.n_mask = nullptr; A.node_cut = nullptr;
    A.n_cut = 0; A.cut_level = 0; A.shard_rank = 0; A.shard_world = 1; A.xbuf = nullptr; A.xbuf_len = 0;
    A.x_callback = nullptr; A.x_ctx = nullptr;
    h->sharded = false;
    return 0;
  }
  const dompc_problem_desc& d = h->d;
  if (!h->shard_capable) { h->error = "this code object was built without tree-sharding support (-DDOMPC_SHARD=1)"; return 1; }
  if (s->world < 1 || s->rank < 0 || s->rank >= s->world || s->cut_level < 1 || s->n_cut < 1) { h->error = "invalid shard description"; return 1; }
  if (!s->x_mask || !s->g_mask || !s->edge_mask || !s->node_mask || !s->node_cut || !s->xbuf) { h->error = "null pointer in shard description"; return 1; }
#ifndef DOMPC_HOST_EMU
  if (!s->allreduce && !h->rccl_comm) { h->error = "no collective: pass allreduce or call dompc_rccl_init first"; return 1; }
#else
  if (!s->allreduce) { h->error = "the host emulation needs an allreduce callback"; return 1; }
#endif
#ifndef DOMPC_HOST_EMU
  HIPCHK(h, hipSetDevice(d.device));
#endif
  if (s->xbuf_doubles < dompc_exchange_doubles(h, s->world, s->n_cut)) { h->error = "exchange buffer too small (dompc_exchange_doubles)"; return 1; }
  int rc = 0;
  rc |= upload(h, &A.x_mask, s->x_mask, d.n_opt_x);
  rc |= upload(h, &A.g_mask, s->g_mask, d.n_g);
  rc |= upload(h, &A.e_mask, s->edge_mask, d.n_edges);
  rc |= upload(h, &A.n_mask, s->node_mask, d.n_nodes);
  rc |= upload(h, &A.node_cut, s->node_cut, d.n_nodes);
  if (rc) return 1;
  A.n_cut = s->n_cut; A.cut_level = s->cut_level; A.shard_rank = s->rank; A.shard_world = s->world;
  A.xbuf = s->xbuf; A.xbuf_len = (int32_t)dompc_exchange_doubles(h, s->world, s->n_cut);
  h->allreduce = s->allreduce; h->allreduce_ctx = s->ctx;
#ifndef DOMPC_HOST_EMU
  if (!h->x_words) {
    HIPCHK(h, hipHostMalloc((void**)&h->x_words, 64, hipHostMallocMapped));
    memset(h->x_words, 0, 64);
  }
  if (!h->shard_stream) {
    // The resident solver kernel of a sharded solve waits for collectives (RCCL kernels on the caller's streams).
    // Streams of different priorities never share a hardware queue, so it runs on its own lowest-priority stream
    // and the collective is never queued behind the kernel that is waiting for it.
    int least = 0, greatest = 0;
    HIPCHK(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIPCHK(h, hipStreamCreateWithPriority(&h->shard_stream, hipStreamNonBlocking, least));
  }
  void* dw = nullptr;
  HIPCHK(h, hipHostGetDevicePointer(&dw, h->x_words, 0));
  A.x_req = (volatile uint32_t*)dw; A.x_ack = (volatile uint32_t*)dw + 1; A.x_count = (volatile uint32_t*)dw + 2;
  A.x_callback = nullptr; A.x_ctx = nullptr;
#else
  A.x_callback = s->allreduce; A.x_ctx = s->ctx;
#endif
  if (dev_sync(h)) return 1;
  h->sharded = true;
  return 0;
}

#ifndef DOMPC_HOST_EMU
// Host side of the exchange handshake (Thr::xchg): serve the kernel's requests until it has finished.
static int serve_exchanges(dompc_handle* h, hipStream_t st) {
  hipEvent_t done;
  HIPCHK(h, hipEventCreateWithFlags(&done, hipEventDisableTiming));
  HIPCHK(h, hipEventRecord(done, st));
  volatile uint32_t* w = h->x_words;
  uint32_t served = 0;
  int rc = 0;
  const auto t0 = std::chrono::steady_clock::now();
  bool raised = false;
  unsigned polls = 0;
  while (true) {
    if ((++polls & 0xfffu) == 0) {               // watchdog (see dev_sync_watchdog)
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (!raised && dt > h->watchdog_s) { int32_t z_ = 0; __atomic_compare_exchange_n(h->abort_word, &z_, 1, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE); raised = true; }   // (0 -> 1: a user's request, 2, is left alone)
      if (raised && dt > h->watchdog_s + 30.0) { h->error = "watchdog: the sharded solve did not finish"; rc = 1; break; }
    }
    const uint32_t r = w[0];
    if (r != served) {
      __sync_synchronize();
      const uint32_t count = w[2], off = w[3];
      if ((int64_t)off + count > h->base.xbuf_len) { h->error = "exchange request outside the buffer"; rc = 1; w[1] = r; served = r; continue; }
      if (h->allreduce) {
        h->allreduce(h->allreduce_ctx, h->base.xbuf + off, (int32_t)count);
      } else {                                   // native RCCL: in-place sum of doubles (ncclFloat64 = 8, ncclSum = 0)
        double* buf = h->base.xbuf + off;
        const int e1 = h->nccl_all_reduce(buf, buf, (size_t)count, 8, 0, h->rccl_comm, h->rccl_stream);
        if (e1 != 0 || hipStreamSynchronize(h->rccl_stream) != hipSuccess) { h->error = "ncclAllReduce failed in the exchange loop"; rc = 1; }
      }
      served = r;
      __sync_synchronize();
      w[1] = r;
    } else {
      const hipError_t q = hipEventQuery(done);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) { h->error = std::string("sharded solve: ") + hipGetErrorString(q); rc = 1; break; }
    }
  }
  hipEventDestroy(done);
  if (raised) { int32_t expect = 1; __atomic_compare_exchange_n(h->abort_word, &expect, 0, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE); }
  h->n_exchanges = (int64_t)served;             // (the kernel numbers its requests 1, 2, ...: the last one served = their count)
  return rc;
}
#endif

extern "C" int64_t dompc_last_exchange_count(const dompc_handle* h) { return h ? h->n_exchanges : 0; }

extern "C" int dompc_edges_per_wavefront(const dompc_handle* h) { return h ? (int)h->edges_per_wave : 0; }
extern "C" int dompc_batch_object_state(const dompc_handle* h) {
  if (!h) return -1;
#ifndef DOMPC_HOST_EMU
  if (h->fn_solve_batch) return 1;
#endif
  return h->batch_object_stale ? 2 : 0;
}

extern "C" int dompc_solve_batch_device(dompc_handle* h, int32_t B, const double* x0, const double* lbx, const double* ubx,
                                        const double* lbg, const double* ubg, const double* p, double* x, double* g,
                                        double* lam_x, double* lam_g, double* f, dompc_stats* stats, void* stream) {
  if (!h) return 1;
  if (B <= 0) return 0;
  if (!x0 || !lbx || !ubx || !lbg || !ubg || !p) { h->error = "null input pointer"; return 1; }
#ifndef DOMPC_HOST_EMU
  HIPCHK(h, hipSetDevice(h->d.device));
#endif
  dompc::KArgs A = h->base;
  A.x0 = x0; A.lbx = lbx; A.ubx = ubx; A.lbg = lbg; A.ubg = ubg; A.p = p;
  A.x_out = x; A.g_out = g; A.lam_x_out = lam_x; A.lam_g_out = lam_g; A.f_out = f; A.stats = stats;
  A.batch = B; A.mode = 0;
  const int block = fit_block(h, h->block_auto ? (B >= BATCH_ONE_WAVE ? 64 : 256) : h->block);
  const int cap = (h->block_auto && block != 64 && h->slots256 < h->n_slots) ? h->slots256 : h->n_slots;   // resident workgroups at this block size
  int grid = B < cap ? B : cap;
  A.wide = 1;
  A.wide_spread = 0;
  if (h->sharded) {
    if (B != 1) { h->error = "a sharded handle solves one problem per call"; return 1; }
#ifndef DOMPC_HOST_EMU
    if (!stream) { h->error = "sharded solve needs a non-blocking stream (not the default stream)"; return 1; }
    h->x_words[0] = h->x_words[1] = h->x_words[2] = h->x_words[3] = 0;
#endif
  }
#ifndef DOMPC_HOST_EMU
  // small batches: several workgroups per problem so that one make_step can use many CUs
  const char* wenv = getenv("DOMPC_WIDE");
  // enough workgroups that every wavefront owns about two edges, fewer when the batch itself fills the chip
  // (measured on MI355X, single cold industrial_poly step, round-2 kernels: K = 4 / 8 / 16 / 32 -> 27.1 / 23.4 / 22.6 / 26.0 ms;
  //  CSTR 9.1 / - / 8.3 / 10.2 ms - the device-scope barriers grow with K faster than the phases shrink)
  int K = wenv ? atoi(wenv) : (B <= 8 ? 16 : (B <= 32 ? 8 : (B <= 64 ? 4 : 1)));
  if (!wenv && K > h->d.n_edges / 8) K = h->d.n_edges / 8 > 1 ? h->d.n_edges / 8 : 1;
  // Whole-chip placement (KArgs::wide_spread): the K workgroups of a problem are consecutive blocks, which the dispatcher spreads over
  // all XCDs, instead of K blocks of ONE XCD (<= 32 CUs); the device-scope barrier then keeps its L2 write-back (xcd_census sees the
  // placement).  One problem alone always runs this way: measured on MI355X (tools/gpu_wide_spread.py, profiles/r05_wide_spread.txt) the
  // 243-leaf tree of BASELINE configs[4] (4 008 edges) takes 72.3 ms with K = 32 on one XCD, 67.4 ms with the same K spread and
  // 49.7 / 48.9 / 49.3 / 55.6 / 63.8 ms with K = 64 / 96 / 128 / 192 / 256; the shipped 9-scenario problem (180 edges) 19.1 ms with 16
  // workgroups of one XCD and 18.2 / 18.0 / 18.8 ms with 16 / 24 / 32 spread - the barriers grow with K, the phases shrink with
  // sqrt-like returns: K ~ 12 sqrt(edges / 45).  DOMPC_WIDE_SPREAD=0/1 and DOMPC_WIDE override.
  const char* senv = getenv("DOMPC_WIDE_SPREAD");
  int auto_block = 256;
  bool spread = senv ? atoi(senv) != 0 : (!h->sharded && (B == 1 || (B <= 4 && h->d.n_edges >= 2048)));
  if (spread && !wenv) {
    // Round 6 (tools/gpu_b1_k_sweep.sh, tools/gpu_tree_k_sweep.sh; the phases got faster, the barriers did not): the 180-edge problem
    // 24.7 / 22.8 / 22.9 / 23.1 / 23.5 / 24.0 / 25.7 ms (84 iterations) with K = 8 / 12 / 16 / 20 / 24 / 32 / 48, the 243-leaf tree 53.8 /
    // 49.5 / 38.9 / 39.5 / 39.6 / 39.8 / 40.1 ms with K = 32 / 48 / 64 / 80 / 96 / 112 / 128 - K ~ 6.4 sqrt(edges / 45) in steps of four,
    // and at least one wavefront per scenario chain (four wavefronts per workgroup): below that some wavefronts walk two chains of the
    // Riccati passes one after the other - the step between K = 48 and 64 on the tree.
    int Ks = 4 * (int)lround(1.6 * sqrt((double)h->d.n_edges / 45.0));
    if (4 * Ks < h->n_leaves) Ks = 4 * ((h->n_leaves + 15) / 16);
    // ... and for the small problems on the four-edge sweep two wavefronts per workgroup instead of four, twice the workgroups
    // (tools/gpu_b1_block_sweep.sh: the 180-edge problem 22.6 ms with 12 x 256 threads, 21.3 ms with 24 x 128 - 48 wavefronts either way,
    // one quad of edges each; 20 x 128: 22.6, 28 x 128: 21.7, 32 x 128: 21.7, 32 x 64: 23.2; the tree prefers 64 x 256: 38.9 against 42.6 ms
    // with 128 x 128 - there the barrier grows with the workgroups)
    if (h->edges_per_wave == 4 && Ks <= 16 && !getenv("DOMPC_WIDE_BLOCK")) { auto_block = 128; Ks *= 2; }
    if (Ks > h->d.n_edges / 7) Ks = h->d.n_edges / 7;
    if (Ks > 256 / B) Ks = 256 / B;
    // (fewer than eight workgroups: the problem is too small for the whole chip - its few workgroups stay on one XCD with the light
    //  barrier; measured: CSTR nominal, 20 edges, 4.9 ms on one XCD against 5.6 ms with its two workgroups on two XCDs)
    if (Ks >= 8 || senv) K = Ks < 1 ? 1 : Ks; else spread = false;
  }
  if (K > (spread ? 256 : 32)) K = spread ? 256 : 32;
  if (spread && (int64_t)B * K > 2048) spread = false;       // (reduction partials: 64 x 32 workgroup rows)
  if (!spread && K > 32) K = 32;
  {
    // Co-residency: the device-scope barrier of the wide mode needs EVERY workgroup of the launch on the chip at the same time - a
    // workgroup that waits for a free CU never arrives and its peers spin until the watchdog.  The grid is therefore capped by what this
    // device keeps resident at the wide mode's workgroup size (occupancy of the kernel x compute units: a partitioned or smaller part,
    // a DOMPC_WIDE / DOMPC_WIDE_SPREAD override); K shrinks until it fits, down to one workgroup per problem.  (What this cannot see:
    // other kernels on the device - INTEGRATION.md section 4.)
    const int wb_ = fit_block(h, 256);
    const int64_t resident = (wb_ == 256 || h->slots64 <= 0) ? h->slots256 : (int64_t)h->slots64 * 64 / wb_;
    auto grid_of = [&](int k, bool sp) -> int64_t { return sp ? (int64_t)B * k : (int64_t)((B + 7) / 8) * 8 * k; };
    while (K > 1 && resident > 0 && grid_of(K, spread) > resident) --K;
  }
  if (K > 1 && B <= 64 && B <= h->n_slots) {
    A.wide = K;
    A.wide_spread = spread ? 1 : 0;
    grid = spread ? B * K : ((B + 7) / 8) * 8 * K;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(h, hipMemsetAsync(A.wide_bar, 0, sizeof(uint32_t) * dompc::WIDE_BAR_STRIDE * 64, st));
    HIPCHK(h, hipMemsetAsync(A.wide_flags, 0, sizeof(int32_t) * 8 * 64, st));
  }
#endif
#ifndef DOMPC_HOST_EMU
  int wide_block = (A.wide > 1 && A.wide_spread) ? auto_block : 256;
#else
  int wide_block = 256;
#endif
  if (const char* wb = getenv("DOMPC_WIDE_BLOCK")) { const int v = atoi(wb); if (v == 64 || v == 128 || v == 256 || v == 512) wide_block = v; }   // (512 needs a code object built with -DDOMPC_MAXBLOCK=512)
  if (launch(h, A, grid, (A.wide > 1 || h->sharded) ? fit_block(h, wide_block) : block, stream)) return 1;
#ifndef DOMPC_HOST_EMU
  if (h->sharded) return serve_exchanges(h, (hipStream_t)stream);
#endif
  return 0;
}

extern "C" int dompc_solve_batch(dompc_handle* h, int32_t B, const double* x0, const double* lbx, const double* ubx,
                                 const double* lbg, const double* ubg, const double* p, double* x, double* g,
                                 double* lam_x, double* lam_g, double* f, dompc_stats* stats) {
  if (!h) return 1;
  if (B <= 0) return 0;
  if (!x0 || !lbx || !ubx || !lbg || !ubg || !p) { h->error = "null input pointer"; return 1; }
  auto t0 = std::chrono::steady_clock::now();
#ifndef DOMPC_HOST_EMU
  HIPCHK(h, hipSetDevice(h->d.device));
#endif
  const dompc_problem_desc& d = h->d;
  if (ensure_staging(h, B)) return 1;
  int rc = 0;
  rc |= h2d(h, h->s_x0, x0, sizeof(double) * (size_t)B * d.n_opt_x);
  rc |= h2d(h, h->s_p, p, sizeof(double) * (size_t)B * d.n_opt_p);
  rc |= h2d(h, h->s_lbx, lbx, sizeof(double) * d.n_opt_x);
  rc |= h2d(h, h->s_ubx, ubx, sizeof(double) * d.n_opt_x);
  rc |= h2d(h, h->s_lbg, lbg, sizeof(double) * d.n_g);
  rc |= h2d(h, h->s_ubg, ubg, sizeof(double) * d.n_g);
  if (rc) return 1;
  static const bool timing = getenv("DOMPC_TIMING") != nullptr;       // measurement aid: where the wall time of a host-buffer call goes (stderr)
  auto t1 = std::chrono::steady_clock::now();
  if (h->sharded && dev_sync(h)) return 1;      // the sharded solve runs on its own stream: inputs must have landed
  if (dompc_solve_batch_device(h, B, h->s_x0, h->s_lbx, h->s_ubx, h->s_lbg, h->s_ubg, h->s_p, h->s_x, h->s_g, h->s_lamx,
                               h->s_lamg, h->s_f, h->s_stats, own_stream(h)))
    return 1;
#ifndef DOMPC_HOST_EMU
  // wait here (bounded by the watchdog) before the result copies are queued behind the kernel
  auto t2 = std::chrono::steady_clock::now();
  if (!h->sharded && dev_sync_watchdog(h, h->stream, (B + h->n_slots - 1) / (h->n_slots > 0 ? h->n_slots : 1))) return 1;
#else
  auto t2 = std::chrono::steady_clock::now();
#endif
  auto t3 = std::chrono::steady_clock::now();
  if (x) rc |= d2h(h, x, h->s_x, sizeof(double) * (size_t)B * d.n_opt_x);
  if (g) rc |= d2h(h, g, h->s_g, sizeof(double) * (size_t)B * d.n_g);
  if (lam_x) rc |= d2h(h, lam_x, h->s_lamx, sizeof(double) * (size_t)B * d.n_opt_x);
  if (lam_g) rc |= d2h(h, lam_g, h->s_lamg, sizeof(double) * (size_t)B * d.n_g);
  if (f) rc |= d2h(h, f, h->s_f, sizeof(double) * (size_t)B);
  if (stats) rc |= d2h(h, stats, h->s_stats, sizeof(dompc_stats) * (size_t)B);
  if (rc || dev_sync(h)) return 1;
  if (timing) {
    auto t4 = std::chrono::steady_clock::now();
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "[dompc timing] B=%d  inputs queued %.0f us | launch call %.0f us | wait for the kernel %.0f us | results copied %.0f us\n",
            (int)B, us(t0, t1), us(t1, t2), us(t2, t3), us(t3, t4));
  }
  if (stats) {
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int b = 0; b < B; ++b) stats[b].t_wall_total = dt;
  }
  return 0;
}

extern "C" int dompc_solve(dompc_handle* h, const double* x0, const double* lbx, const double* ubx, const double* lbg,
                           const double* ubg, const double* p, const double* /*lam_x0*/, const double* /*lam_g0*/,
                           double* x, double* g, double* lam_x, double* lam_g, double* f, dompc_stats* stats) {
  return dompc_solve_batch(h, 1, x0, lbx, ubx, lbg, ubg, p, x, g, lam_x, lam_g, f, stats);
}

extern "C" int dompc_sweep_batch_device(dompc_handle* h, int32_t B, const double* x, const double* lam, const double* p,
                                        double* g, double* blocks, void* stream) {
  if (!h) return 1;
  if (B <= 0) return 0;
  if (!x || !lam || !p || !g) { h->error = "null pointer"; return 1; }      // (blocks may be null: residuals only - the timing of the sweep itself, bench.py)
#ifndef DOMPC_HOST_EMU
  HIPCHK(h, hipSetDevice(h->d.device));
#endif
  dompc::KArgs A = h->base;
  A.p = p; A.sw_x = x; A.sw_lam = lam; A.sw_g = g; A.sw_blocks = blocks;
  A.batch = B; A.mode = 2;
  // (same launch shape as a solve of that many problems: one 64-thread workgroup per iterate from BATCH_ONE_WAVE on)
  const int block = fit_block(h, h->block_auto ? (B >= BATCH_ONE_WAVE ? 64 : 256) : h->block);
  const int cap = (h->block_auto && block != 64 && h->slots256 < h->n_slots) ? h->slots256 : h->n_slots;
  const int grid = B < cap ? B : cap;
  return launch(h, A, grid, block, stream);
}

static int newton_step_impl(dompc_handle* h, const double* x, const double* lam_g, const double* zl,
                            const double* zu, const double* lbx, const double* ubx, const double* lbg,
                            const double* ubg, const double* p, double mu, double delta_w, double* dx,
                            double* dlam, double* rd, double* c, int at_solution) {
  if (!h) return 1;
#ifndef DOMPC_HOST_EMU
  HIPCHK(h, hipSetDevice(h->d.device));
#endif
  const dompc_problem_desc& d = h->d;
  if (h->sharded) { h->error = "dompc_debug_newton_step is not available on a sharded handle (call dompc_set_sharding(h, NULL) first)"; return 1; }
  if (ensure_staging(h, 1)) return 1;
  int rc = 0;
  rc |= h2d(h, h->s_x0, x, sizeof(double) * d.n_opt_x);
  rc |= h2d(h, h->s_p, p, sizeof(double) * d.n_opt_p);
  rc |= h2d(h, h->s_lbx, lbx, sizeof(double) * d.n_opt_x);
  rc |= h2d(h, h->s_ubx, ubx, sizeof(double) * d.n_opt_x);
  rc |= h2d(h, h->s_lbg, lbg, sizeof(double) * d.n_g);
  rc |= h2d(h, h->s_ubg, ubg, sizeof(double) * d.n_g);
  rc |= h2d(h, h->s_dbg[0], lam_g, sizeof(double) * d.n_g);
  rc |= h2d(h, h->s_dbg[1], zl, sizeof(double) * d.n_opt_x);
  rc |= h2d(h, h->s_dbg[2], zu, sizeof(double) * d.n_opt_x);
  if (rc) return 1;
  dompc::KArgs A = h->base;
  A.x0 = h->s_x0; A.lbx = h->s_lbx; A.ubx = h->s_ubx; A.lbg = h->s_lbg; A.ubg = h->s_ubg; A.p = h->s_p;
  A.dbg_lam = h->s_dbg[0]; A.dbg_zl = h->s_dbg[1]; A.dbg_zu = h->s_dbg[2];
  A.dbg_dx = h->s_dbg[3]; A.dbg_dlam = h->s_dbg[4]; A.dbg_rd = h->s_dbg[5]; A.dbg_c = h->s_dbg[6];
  A.dbg_mu = mu; A.dbg_delta = delta_w;
  A.batch = 1; A.mode = 1; A.dbg_at_solution = at_solution;
  if (launch(h, A, 1, fit_block(h, 256), main_stream(h))) return 1;
  if (dx) rc |= d2h(h, dx, h->s_dbg[3], sizeof(double) * d.n_opt_x);
  if (dlam) rc |= d2h(h, dlam, h->s_dbg[4], sizeof(double) * d.n_g);
  if (rd) rc |= d2h(h, rd, h->s_dbg[5], sizeof(double) * d.n_opt_x);
  if (c) rc |= d2h(h, c, h->s_dbg[6], sizeof(double) * d.n_g);
  if (rc || dev_sync(h)) return 1;
  return 0;
}

extern "C" int dompc_debug_newton_step(dompc_handle* h, const double* x, const double* lam_g, const double* zl,
                                       const double* zu, const double* lbx, const double* ubx, const double* lbg,
                                       const double* ubg, const double* p, double mu, double delta_w, double* dx,
                                       double* dlam, double* rd, double* c) {
  return newton_step_impl(h, x, lam_g, zl, zu, lbx, ubx, lbg, ubg, p, mu, delta_w, dx, dlam, rd, c, 0);
}
extern "C" int dompc_newton_step_at_solution(dompc_handle* h, const double* x, const double* lam_g, const double* zl,
                                             const double* zu, const double* lbx, const double* ubx, const double* lbg,
                                             const double* ubg, const double* p, double mu, double* dx, double* dlam) {
  return newton_step_impl(h, x, lam_g, zl, zu, lbx, ubx, lbg, ubg, p, mu, 0.0, dx, dlam, nullptr, nullptr, 1);
}

// B parameter vectors at ONE point (dompc_ipm.h): one launch per chunk of resident slots, one workgroup per vector
extern "C" int dompc_newton_steps_at_solution(dompc_handle* h, int32_t B, const double* x, const double* lam_g, const double* zl,
                                              const double* zu, const double* lbx, const double* ubx, const double* lbg,
                                              const double* ubg, const double* p, double mu, double* dx, double* dlam) {
  if (!h || B < 1) return 1;
#ifndef DOMPC_HOST_EMU
  HIPCHK(h, hipSetDevice(h->d.device));
#endif
  const dompc_problem_desc& d = h->d;
  if (h->sharded) { h->error = "dompc_newton_steps_at_solution is not available on a sharded handle"; return 1; }
  int chunk = h->n_slots < B ? h->n_slots : B;
#ifdef DOMPC_HOST_EMU
  chunk = B;
#endif
  if (chunk < 1) chunk = 1;
  if (ensure_staging(h, chunk)) return 1;
  int rc = 0;
  rc |= h2d(h, h->s_x0, x, sizeof(double) * d.n_opt_x);
  rc |= h2d(h, h->s_lbx, lbx, sizeof(double) * d.n_opt_x);
  rc |= h2d(h, h->s_ubx, ubx, sizeof(double) * d.n_opt_x);
  rc |= h2d(h, h->s_lbg, lbg, sizeof(double) * d.n_g);
  rc |= h2d(h, h->s_ubg, ubg, sizeof(double) * d.n_g);
  rc |= h2d(h, h->s_dbg[0], lam_g, sizeof(double) * d.n_g);
  rc |= h2d(h, h->s_dbg[1], zl, sizeof(double) * d.n_opt_x);
  rc |= h2d(h, h->s_dbg[2], zu, sizeof(double) * d.n_opt_x);
  if (rc) return 1;
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = (B - b0 < chunk) ? B - b0 : chunk;
    if (h2d(h, h->s_p, p + (size_t)b0 * d.n_opt_p, sizeof(double) * (size_t)nb * d.n_opt_p)) return 1;
    dompc::KArgs A = h->base;
    A.x0 = h->s_x0; A.lbx = h->s_lbx; A.ubx = h->s_ubx; A.lbg = h->s_lbg; A.ubg = h->s_ubg; A.p = h->s_p;
    A.dbg_lam = h->s_dbg[0]; A.dbg_zl = h->s_dbg[1]; A.dbg_zu = h->s_dbg[2];
    A.dbg_dx = h->s_x; A.dbg_dlam = h->s_lamg; A.dbg_rd = h->s_lamx; A.dbg_c = h->s_g;        // (batch staging buffers: nb rows each)
    A.dbg_mu = mu; A.dbg_delta = 0.0;
    A.batch = nb; A.mode = 1; A.dbg_at_solution = 1;
    if (launch(h, A, nb, fit_block(h, 256), main_stream(h))) return 1;
    rc |= d2h(h, dx + (size_t)b0 * d.n_opt_x, h->s_x, sizeof(double) * (size_t)nb * d.n_opt_x);
    rc |= d2h(h, dlam + (size_t)b0 * d.n_g, h->s_lamg, sizeof(double) * (size_t)nb * d.n_g);
    if (rc || dev_sync(h)) return 1;
  }
  return 0;
}

// Iteration trace of problem 0 of the last solve call: rows of 8 doubles
// (it, mu, E0, inf_pr, inf_du, +-alpha (negative = line search failed), delta_w, obj).
extern "C" int dompc_debug_get_trace(dompc_handle* h, double* out, int32_t max_rows) {
  if (!h || !out) return 1;
  int rows = max_rows < h->trace_cap ? max_rows : h->trace_cap;
  if (d2h(h, out, h->s_trace, sizeof(double) * 8 * (size_t)rows) || dev_sync(h)) return 1;
  return 0;
}
