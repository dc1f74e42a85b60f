// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for this project's access pattern: 8 B per lane (f64),
// coalesced, far beyond the 256 MiB Infinity Cache (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte
// count in your own access pattern").  Reads N doubles and writes N doubles per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void stream_f64(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = in[i] * 2.0 + 1.0;
}
int main(int argc, char** argv) {
  const size_t n = (argc > 1 ? (size_t)atoll(argv[1]) : (size_t)1 << 28);      // 2 GiB in, 2 GiB out
  double *a, *b;
  if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&b, n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 0, n * 8);
  hipMemset(b, 0, n * 8);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) stream_f64<<<2048, 256>>>(a, b, n);
  hipDeviceSynchronize();
  printf("stream_f64: %zu doubles read and %zu doubles written per launch = %.1f MB each\n", n, n, n * 8 / 1e6);
  return 0;
}
