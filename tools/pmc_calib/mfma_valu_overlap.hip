// Micro-benchmark (measurement aid, gfx950): do FP64 matrix-core instructions and FP64 vector FMAs of the SAME SIMD overlap?
// A workgroup has 8 wavefronts = 2 per SIMD.  Modes:
//   0: every wavefront runs the MFMA stream            1: every wavefront runs the vector-FMA stream
//   2: every wavefront runs both streams interleaved   3: wavefronts 0-3 MFMA, 4-7 vector FMA (one of each per SIMD)
//   4: wavefronts 0-3 MFMA, 4-7 idle                   5: wavefronts 0-3 idle, 4-7 vector FMA
// Each stream: NI independent chains (no dependent-issue stalls), REP trips.  Prints cycles per instruction per wavefront.
//   hipcc --offload-arch=gfx950 -O3 tools/pmc_calib/mfma_valu_overlap.hip -o tools/pmc_calib/mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NM = 4;          // independent accumulator tiles
constexpr int NF = 16;         // independent FMA chains

template <int MODE, int OP>
__global__ __launch_bounds__(512) void k(double* out, long long* cyc, int rep, double a, double b) {
  const int w = threadIdx.x >> 6;
  const bool do_m = MODE == 0 || MODE == 2 || ((MODE == 3 || MODE == 4) && w < 4);
  const bool do_f = MODE == 1 || MODE == 2 || ((MODE == 3 || MODE == 5) && w >= 4);
  d4 acc[NM];
  double f[NF];
  float g[NF];
  unsigned u[NF];
  const float af = (float)a, bf = (float)b;
  const unsigned au = (unsigned)(a * 3.0) | 1u, bu = (unsigned)rep;
  for (int i = 0; i < NM; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  for (int i = 0; i < NF; ++i) { f[i] = threadIdx.x * 1e-9 + i; g[i] = (float)f[i]; u[i] = threadIdx.x + i; }
  auto fill = [&](int i) {
    if (OP == 0) f[i] = __builtin_fma(f[i], a, b);
    else if (OP == 1) g[i] = __builtin_fmaf(g[i], af, bf);
    else u[i] = (u[i] ^ au) + bu;                 // (v_xad_u32: one full-rate 32-bit instruction, like index / select arithmetic)
  };
  __syncthreads();
  const long long t0 = clock64();
  if (do_m && do_f) {
    for (int r = 0; r < rep; ++r) {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NF / NM; ++q) fill(i * (NF / NM) + q);
      }
    }
  } else if (do_m) {
    for (int r = 0; r < rep; ++r)
#pragma unroll
      for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  } else if (do_f) {
    for (int r = 0; r < rep; ++r)
#pragma unroll
      for (int i = 0; i < NF; ++i) fill(i);
  }
  const long long t1 = clock64();
  double s = 0.0;
  for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NF; ++i) s += f[i] + (double)g[i] + (double)u[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

template <int MODE, int OP>
void run(const char* what, int grid, int rep, double* out, long long* cyc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, OP>), dim3(grid), dim3(512), 0, 0, out, cyc, rep, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, OP>), dim3(grid), dim3(512), 0, 0, out, cyc, rep, 1.0000001, 1e-9);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(grid * 8);
  hipMemcpy(h.data(), cyc, sizeof(long long) * grid * 8, hipMemcpyDeviceToHost);
  double cm = 0, cf = 0;
  for (int g = 0; g < grid; ++g)
    for (int w = 0; w < 8; ++w) (w < 4 ? cm : cf) += (double)h[g * 8 + w];
  cm /= grid * 4.0; cf /= grid * 4.0;
  printf("%s mode %d  %-46s  %8.3f ms   clock64 ticks per trip: waves 0-3 %9.2f  waves 4-7 %9.2f   (trip = %d MFMA and / or %d filler instructions)\n", OP == 0 ? "v_fma_f64" : (OP == 1 ? "v_fma_f32" : "v_xad_u32"), MODE, what, ms,
         cm / rep, cf / rep, NM, NF);
}

int main() {
  const int grid = 256, rep = 20000;
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * grid * 512);
  hipMalloc(&cyc, sizeof(long long) * grid * 8);
  run<0, 0>("all 8 wavefronts: MFMA f64 16x16x4", grid, rep, out, cyc);
  run<1, 0>("all 8 wavefronts: filler", grid, rep, out, cyc);
  run<2, 0>("all 8 wavefronts: both, interleaved", grid, rep, out, cyc);
  run<3, 0>("waves 0-3 MFMA, waves 4-7 filler", grid, rep, out, cyc);
  run<4, 0>("waves 0-3 MFMA, waves 4-7 idle", grid, rep, out, cyc);
  run<5, 0>("waves 0-3 idle, waves 4-7 filler", grid, rep, out, cyc);
  // the same with single-precision and 32-bit integer fillers: does index / select arithmetic hide under FP64 matrix instructions?
  run<1, 1>("all 8 wavefronts: filler", grid, rep, out, cyc);
  run<2, 1>("all 8 wavefronts: both, interleaved", grid, rep, out, cyc);
  run<3, 1>("waves 0-3 MFMA, waves 4-7 filler", grid, rep, out, cyc);
  run<5, 1>("waves 0-3 idle, waves 4-7 filler", grid, rep, out, cyc);
  run<1, 2>("all 8 wavefronts: filler", grid, rep, out, cyc);
  run<2, 2>("all 8 wavefronts: both, interleaved", grid, rep, out, cyc);
  run<3, 2>("waves 0-3 MFMA, waves 4-7 filler", grid, rep, out, cyc);
  run<5, 2>("waves 0-3 idle, waves 4-7 filler", grid, rep, out, cyc);
  return 0;
}
