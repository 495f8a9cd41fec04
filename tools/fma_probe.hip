// How long does a dependent fp64 FMA take on a SIMD that runs ONE wave (ts_schedule / ts_resident: 256-thread
// workgroups, one per CU)?  C independent chains interleaved, 4 waves per CU as in those kernels.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/fma_probe tools/fma_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)

template <int C>
__global__ __launch_bounds__(256, 1) void chains(double *out, int iters, double a, double b) {
  double x[C];
#pragma unroll
  for (int c = 0; c < C; ++c) x[c] = (double)(threadIdx.x + c) * 1e-3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < C; ++c) x[c] = fma(x[c], a, b);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < C; ++c) s += x[c];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int C>
int run(double *out, int waves_per_simd) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 20000 / C;
  const int grid = 256 * waves_per_simd;
  hipLaunchKernelGGL(chains<C>, dim3(grid), dim3(256), 0, 0, out, iters, 0.999, 1e-3);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(chains<C>, dim3(grid), dim3(256), 0, 0, out, iters, 0.999, 1e-3);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double fmas = (double)iters * 16 * C * waves_per_simd;  // per SIMD
  printf("%d chain(s), %d wave(s) per SIMD: %.3f ns per wave-FMA on a SIMD (%.2f TFLOP/s fp64 on 256 CUs)\n", C, waves_per_simd,
         ms * 1e6 / fmas, fmas * 64 * 2 * 1024 / (ms * 1e-3) / 1e12);
  return 0;
}

int main() {
  double *out;
  CK(hipMalloc(&out, 512 * 256 * sizeof(double)));
  for (int w = 1; w <= 2; ++w) {
    if (run<1>(out, w)) return 1;
    if (run<2>(out, w)) return 1;
    if (run<4>(out, w)) return 1;
    if (run<8>(out, w)) return 1;
  }
  return 0;
}
