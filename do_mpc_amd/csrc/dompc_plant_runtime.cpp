// dompc_plant_runtime.cpp - host side of the batched plant integrator behind the C ABI of include/dompc_ipm.h
// (dompc_plant_*).  Generic: sizes come from the per-model code object (dompc_plant_info_kernel).
// Build flavours as dompc_runtime.cpp: product = part of libdompc_ipm.so (HIP only); test = g++ -DDOMPC_HOST_EMU
// together with dompc_plant.hip compiled as C++ (tests/_hostemu; never shipped).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dompc_ipm.h"
#include "dompc_plant_args.h"

#ifndef DOMPC_HOST_EMU
#include <hip/hip_runtime.h>
#else
extern "C" void dompc_plant_hostemu_info(int64_t* out, char* hash);
extern "C" void dompc_plant_hostemu_run(const dompc_plantk::Args* A);
#endif

static thread_local std::string g_plant_create_error;

struct dompc_plant {
  dompc_plant_desc d;
  std::string error;
  int32_t cap = 0;
  double *s_x = nullptr, *s_u = nullptr, *s_tvp = nullptr, *s_p = nullptr, *s_w = nullptr, *s_v = nullptr, *s_xn = nullptr, *s_y = nullptr;
  int32_t* s_st = nullptr;
  std::vector<void*> allocs;
  int32_t nz = 0;                       // algebraic states of the model (from the code object)
  int32_t method = 0, explicit_limit = 4000;
  std::vector<double> z0;               // Newton start of the algebraic states (dompc_plant_set_z0; zeros by default)
  double* z_dev = nullptr;              // [z_cap][nz]: per-sample start values, carried from call to call
  int32_t z_cap = 0, z_alloc = 0;       // rows with valid start values / rows allocated
  bool z_carry_host = false;            // host entry (dompc_plant_step_batch): rows of consecutive calls are the same trajectories (dompc_plant_set_z_carry)
  bool z_reseed_next = false;
#ifndef DOMPC_HOST_EMU
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr, fn_info = nullptr;
  hipStream_t stream = nullptr;
#endif
};

#ifndef DOMPC_HOST_EMU
#define PHIP(h, expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) { (h)->error = std::string(#expr) + ": " + hipGetErrorString(_e); return 1; } \
  } while (0)
static int palloc(dompc_plant* h, void** p, size_t bytes) {
  PHIP(h, hipMalloc(p, bytes ? bytes : 8));
  h->allocs.push_back(*p);
  return 0;
}
static void pfree(void* p) { (void)hipFree(p); }
#else
static int palloc(dompc_plant* h, void** p, size_t bytes) {
  *p = calloc(1, bytes ? bytes : 8);
  if (!*p) { h->error = "out of memory"; return 1; }
  h->allocs.push_back(*p);
  return 0;
}
static void pfree(void* p) { free(p); }
#endif

extern "C" const char* dompc_plant_last_error(const dompc_plant* h) { return h ? h->error.c_str() : g_plant_create_error.c_str(); }

extern "C" void dompc_plant_destroy(dompc_plant* h) {
  if (!h) return;
#ifndef DOMPC_HOST_EMU
  (void)hipSetDevice(h->d.device);
#endif
  for (void* p : h->allocs) pfree(p);
#ifndef DOMPC_HOST_EMU
  if (h->module) (void)hipModuleUnload(h->module);
  if (h->stream) (void)hipStreamDestroy(h->stream);
#endif
  delete h;
}

extern "C" int dompc_plant_create(const dompc_plant_desc* desc, dompc_plant** out) {
  if (!desc || !out) { g_plant_create_error = "null argument"; return 1; }
  dompc_plant* h = new dompc_plant();
  h->d = *desc;
  auto fail = [&]() { g_plant_create_error = h->error; dompc_plant_destroy(h); *out = nullptr; return 1; };
  if (desc->nx <= 0) { h->error = "plant without states"; return fail(); }
  if (!(desc->t_step > 0.0) && !desc->discrete) { h->error = "t_step must be positive"; return fail(); }
  int64_t info[16] = {0};
  char hash[64] = {0};
#ifndef DOMPC_HOST_EMU
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    h->error = "no HIP device available: the dompc plant integrator requires an AMD GPU (gfx950)";
    return fail();
  }
  if (hipSetDevice(desc->device) != hipSuccess) { h->error = "hipSetDevice failed"; return fail(); }
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { h->error = "hipStreamCreate failed"; return fail(); }
  if (!desc->code_object_path || hipModuleLoad(&h->module, desc->code_object_path) != hipSuccess) {
    h->error = std::string("hipModuleLoad failed for ") + (desc->code_object_path ? desc->code_object_path : "(null)");
    return fail();
  }
  if (hipModuleGetFunction(&h->fn, h->module, "dompc_plant_kernel") != hipSuccess ||
      hipModuleGetFunction(&h->fn_info, h->module, "dompc_plant_info_kernel") != hipSuccess) {
    h->error = "code object lacks the plant kernels"; return fail();
  }
  {
    int64_t* out_d; char* hash_d;
    if (palloc(h, (void**)&out_d, sizeof(info)) || palloc(h, (void**)&hash_d, sizeof(hash))) return fail();
    struct { int64_t* a; char* b; } args = {out_d, hash_d};
    size_t sz = sizeof(args);
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    if (hipModuleLaunchKernel(h->fn_info, 1, 1, 1, 64, 1, 1, 0, h->stream, nullptr, cfg) != hipSuccess ||
        hipMemcpyAsync(info, out_d, sizeof(info), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipMemcpyAsync(hash, hash_d, sizeof(hash), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) {
      h->error = "dompc_plant_info_kernel failed"; return fail();
    }
  }
#else
  dompc_plant_hostemu_info(info, hash);
#endif
  const int64_t want[8] = {desc->nx, desc->nu, desc->np, desc->ntvp, desc->nw, desc->nv, desc->ny, desc->discrete ? 1 : 0};
  for (int i = 0; i < 8; ++i)
    if (info[i] != want[i]) {
      char buf[200];
      snprintf(buf, sizeof(buf), "plant code object was built for different model dimensions (field %d: %lld vs %lld)", i,
               (long long)info[i], (long long)want[i]);
      h->error = buf;
      return fail();
    }
  if (info[8] != (int64_t)sizeof(dompc_plantk::Args)) { h->error = "plant argument layout mismatch between runtime and code object"; return fail(); }
  if (desc->model_hash && strncmp(desc->model_hash, hash, 63) != 0) { h->error = "plant model hash mismatch"; return fail(); }
  h->d.code_object_path = nullptr; h->d.model_hash = nullptr;
  h->nz = (int32_t)info[9];
  h->z0.assign((size_t)(h->nz > 0 ? h->nz : 0), 0.0);
  *out = h;
  return 0;
}

extern "C" int dompc_plant_set_method(dompc_plant* h, int32_t method, int32_t explicit_limit) {
  if (!h) return 1;
  if (method < 0 || method > 2) { h->error = "method must be 0 (explicit, implicit for stiff samples), 1 (explicit) or 2 (implicit)"; return 1; }
  h->method = method;
  if (explicit_limit > 0) h->explicit_limit = explicit_limit;
  return 0;
}

extern "C" int32_t dompc_plant_num_alg_states(const dompc_plant* h) { return h ? h->nz : 0; }

extern "C" int dompc_plant_set_z0(dompc_plant* h, const double* z0) {
  if (!h) return 1;
  for (int i = 0; i < h->nz; ++i) h->z0[(size_t)i] = z0 ? z0[i] : 0.0;
  h->z_cap = 0;                          // (the per-sample values are re-seeded by the next call)
  return 0;
}

// per-sample start values of the algebraic states for B samples: seeded with z0 when the buffer is (re)created, when it grows, and on
// request (`reseed`).  Row b of the buffer belongs to sample b of the CALLS: carried values only make sense when the rows of consecutive
// calls are the same trajectories (the device closed loops, Simulator.make_step); a host call on unrelated samples reseeds.
static int ensure_z(dompc_plant* h, int32_t B, bool reseed) {
  if (h->nz <= 0) return 0;
  if (B <= h->z_cap && !reseed) return 0;
  if (B > h->z_alloc) {
    if (h->z_dev) {
      for (size_t i = 0; i < h->allocs.size(); ++i)
        if (h->allocs[i] == h->z_dev) { h->allocs.erase(h->allocs.begin() + i); pfree(h->z_dev); break; }
      h->z_dev = nullptr;
    }
    h->z_cap = 0; h->z_alloc = 0;
    if (palloc(h, (void**)&h->z_dev, sizeof(double) * (size_t)B * h->nz)) return 1;
    h->z_alloc = B;
  }
  std::vector<double> seed((size_t)B * h->nz);
  for (int32_t b = 0; b < B; ++b)
    for (int i = 0; i < h->nz; ++i) seed[(size_t)b * h->nz + i] = h->z0[(size_t)i];
#ifndef DOMPC_HOST_EMU
  PHIP(h, hipMemcpy(h->z_dev, seed.data(), seed.size() * sizeof(double), hipMemcpyHostToDevice));
#else
  memcpy(h->z_dev, seed.data(), seed.size() * sizeof(double));
#endif
  if (B > h->z_cap) h->z_cap = B;        // a reseed of the first B rows leaves the carried rows behind them alone
  return 0;
}

extern "C" int dompc_plant_set_z_carry(dompc_plant* h, int32_t on) {
  if (!h) return 1;
  h->z_carry_host = on != 0;
  return 0;
}

static int launch_plant(dompc_plant* h, dompc_plantk::Args& A, void* stream) {
#ifndef DOMPC_HOST_EMU
  size_t sz = sizeof(A);
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  PHIP(h, hipModuleLaunchKernel(h->fn, (A.batch + 63) / 64, 1, 1, 64, 1, 1, 0, (hipStream_t)stream, nullptr, cfg));
#else
  (void)h; (void)stream;
  dompc_plant_hostemu_run(&A);
#endif
  return 0;
}

static void fill_args(const dompc_plant* h, dompc_plantk::Args& A, int32_t B, int32_t shared_mask) {
  const dompc_plant_desc& d = h->d;
  A.batch = B;
  A.stride_u = (shared_mask & 1) ? 0 : d.nu; A.stride_tvp = (shared_mask & 2) ? 0 : d.ntvp; A.stride_p = (shared_mask & 4) ? 0 : d.np;
  A.stride_w = (shared_mask & 8) ? 0 : d.nw; A.stride_v = (shared_mask & 16) ? 0 : d.nv;
  A.max_steps = d.max_steps > 0 ? d.max_steps : 200000; A.pad = 0;
  A.method = h->method; A.explicit_limit = h->explicit_limit;
  A.z_guess = (h->nz > 0 && B <= h->z_cap) ? h->z_dev : nullptr;
  A.t_step = d.t_step; A.rtol = d.reltol > 0 ? d.reltol : 1e-10; A.atol = d.abstol > 0 ? d.abstol : 1e-10;
}

extern "C" int dompc_plant_step_batch_device(dompc_plant* h, int32_t B, const double* x, const double* u, const double* tvp,
                                             const double* p, const double* w, const double* v, int32_t shared_mask,
                                             double* x_next, double* y, int32_t* status, void* stream) {
  if (!h) return 1;
  if (B <= 0) return 0;
  const dompc_plant_desc& d = h->d;
  if (!x || !x_next || (d.nu && !u) || (d.ntvp && !tvp) || (d.np && !p)) { h->error = "null pointer"; return 1; }
#ifndef DOMPC_HOST_EMU
  PHIP(h, hipSetDevice(d.device));
#endif
  if (ensure_z(h, B, h->z_reseed_next)) return 1;
  h->z_reseed_next = false;
  dompc_plantk::Args A;
  memset(&A, 0, sizeof(A));
  A.x = x; A.u = u; A.tvp = tvp; A.p = p; A.w = w; A.v = v; A.x_next = x_next; A.y = y; A.status = status;
  fill_args(h, A, B, shared_mask);
  return launch_plant(h, A, stream);
}

extern "C" int dompc_plant_step_batch(dompc_plant* h, int32_t B, const double* x, const double* u, const double* tvp,
                                      const double* p, const double* w, const double* v, int32_t shared_mask,
                                      double* x_next, double* y, int32_t* status) {
  if (!h) return 1;
  if (B <= 0) return 0;
  const dompc_plant_desc& d = h->d;
  if (!x || !x_next || (d.nu && !u) || (d.ntvp && !tvp) || (d.np && !p)) { h->error = "null pointer"; return 1; }
#ifndef DOMPC_HOST_EMU
  PHIP(h, hipSetDevice(d.device));
#endif
  if (B > h->cap) {
    //  the staging buffers are invalid from here until ALL new ones exist: capacity 0 and null pointers first, so
    // that a failed allocation cannot leave a later call (B <= old capacity) copying into freed device memory
    h->cap = 0;
    void** slots[] = {(void**)&h->s_x, (void**)&h->s_u, (void**)&h->s_tvp, (void**)&h->s_p, (void**)&h->s_w, (void**)&h->s_v,
                      (void**)&h->s_xn, (void**)&h->s_y, (void**)&h->s_st};
    for (void** sp : slots) {
      void* q = *sp;
      *sp = nullptr;
      if (q) {
        for (size_t i = 0; i < h->allocs.size(); ++i)
          if (h->allocs[i] == q) { h->allocs.erase(h->allocs.begin() + i); pfree(q); break; }
      }
    }
    const size_t n = (size_t)B * sizeof(double);
    if (palloc(h, (void**)&h->s_x, n * d.nx) || palloc(h, (void**)&h->s_u, n * d.nu) || palloc(h, (void**)&h->s_tvp, n * d.ntvp) ||
        palloc(h, (void**)&h->s_p, n * d.np) || palloc(h, (void**)&h->s_w, n * d.nw) || palloc(h, (void**)&h->s_v, n * d.nv) ||
        palloc(h, (void**)&h->s_xn, n * d.nx) || palloc(h, (void**)&h->s_y, n * d.ny) ||
        palloc(h, (void**)&h->s_st, (size_t)B * sizeof(int32_t)))
      return 1;
    h->cap = B;
  }
  auto rows = [&](int bit) { return (shared_mask & bit) ? (size_t)1 : (size_t)B; };
#ifndef DOMPC_HOST_EMU
  auto up = [&](void* dst, const void* src, size_t bytes) -> int {
    if (!bytes || !src) return 0;
    PHIP(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    return 0;
  };
  auto down = [&](void* dst, const void* src, size_t bytes) -> int {
    if (!bytes || !dst) return 0;
    PHIP(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    return 0;
  };
  void* st = (void*)h->stream;
#else
  auto up = [&](void* dst, const void* src, size_t bytes) -> int { if (bytes && src) memcpy(dst, src, bytes); return 0; };
  auto down = [&](void* dst, const void* src, size_t bytes) -> int { if (bytes && dst) memcpy(dst, src, bytes); return 0; };
  void* st = nullptr;
#endif
  const size_t D = sizeof(double);
  if (up(h->s_x, x, D * B * d.nx) || up(h->s_u, u, D * rows(1) * d.nu) || up(h->s_tvp, tvp, D * rows(2) * d.ntvp) ||
      up(h->s_p, p, D * rows(4) * d.np) || up(h->s_w, w, D * rows(8) * d.nw) || up(h->s_v, v, D * rows(16) * d.nv))
    return 1;
  h->z_reseed_next = !h->z_carry_host;       // unrelated samples per call unless the caller says the rows are the same trajectories
  if (dompc_plant_step_batch_device(h, B, h->s_x, h->s_u, h->s_tvp, h->s_p, w ? h->s_w : nullptr, v ? h->s_v : nullptr, shared_mask,
                                    h->s_xn, y ? h->s_y : nullptr, h->s_st, st))
    return 1;
  if (down(x_next, h->s_xn, D * B * d.nx) || down(y, h->s_y, D * B * d.ny) || down(status, h->s_st, sizeof(int32_t) * (size_t)B)) return 1;
#ifndef DOMPC_HOST_EMU
  PHIP(h, hipStreamSynchronize(h->stream));
#endif
  return 0;
}
