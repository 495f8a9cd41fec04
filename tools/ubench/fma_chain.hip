// Dependent-issue cost of v_fma_f64 on gfx950 with ONE wave per SIMD (the resident kernels' regime): cycles per FMA for
// 1, 2, 3, 4 and 8 independent chains issued round-robin.   hipcc --offload-arch=gfx950 -O3 -o fma_chain fma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CH>
__global__ __launch_bounds__(256, 1) void chain(double *out, long long *cyc, double x, double c) {
  double p[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) p[i] = x + i + threadIdx.x;
  constexpr int kIters = 4096 / CH;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < CH; ++i) p[i] = __builtin_fma(p[i], x, c);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) s += p[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int CH>
void run(double *out, long long *cyc, int grid) {
  long long h = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(chain<CH>, dim3(grid), dim3(256), 0, 0, out, cyc, 0.999999, 1e-9);
    hipDeviceSynchronize();
  }
  hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
  printf("grid %3d  %d chain(s): %.2f cycles (s_memtime ticks) per v_fma_f64 of one wave\n", grid, CH, (double)h / (4096 / CH * 8 * CH));
}

int main() {
  double *out;
  long long *cyc;
  hipMalloc(&out, 256 * 256 * sizeof(double));
  hipMalloc(&cyc, sizeof(long long));
  for (int grid : {1, 256}) {
    run<1>(out, cyc, grid);
    run<2>(out, cyc, grid);
    run<3>(out, cyc, grid);
    run<4>(out, cyc, grid);
    run<8>(out, cyc, grid);
  }
  return 0;
}
