// Accuracy of v_rcp_f64 / v_rsq_f64 on gfx950 and of the Newton steps on top (fast_rcp / fast_rsqrt in csrc/tsamd_device.h):
// maximum relative error against the correctly rounded result, over 2^24 arguments spread over 60 binades.
//   hipcc --offload-arch=gfx950 -O3 -o rcp_accuracy rcp_accuracy.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

__global__ void probe(double *err /* [6] */, unsigned long long seed) {
  unsigned long long s = seed + (blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
  double e[6] = {0, 0, 0, 0, 0, 0};
  for (int it = 0; it < 64; ++it) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const double m = 1.0 + (double)(s >> 11) * (1.0 / 9007199254740992.0);  // [1, 2)
    const int ex = (int)((s >> 3) % 60u) - 30;
    const double x = ldexp(m, ex);
    const double exact = 1.0 / x;  // IEEE division (correctly rounded)
    double r = __builtin_amdgcn_rcp(x);
    e[0] = fmax(e[0], fabs(r - exact) / exact);
    double d = fma(-x, r, 1.0);
    r = fma(r, d, r);
    e[1] = fmax(e[1], fabs(r - exact) / exact);
    d = fma(-x, r, 1.0);
    r = fma(r, d, r);
    e[2] = fmax(e[2], fabs(r - exact) / exact);
    const double exq = 1.0 / sqrt(x);  // (two roundings: good to ~1 ulp itself)
    double y = __builtin_amdgcn_rsq(x);
    e[3] = fmax(e[3], fabs(y - exq) / exq);
    double h = fma(-(x * y), y, 1.0);
    y = fma(0.5 * y, h, y);
    e[4] = fmax(e[4], fabs(y - exq) / exq);
    h = fma(-(x * y), y, 1.0);
    y = fma(0.5 * y, h, y);
    e[5] = fmax(e[5], fabs(y - exq) / exq);
  }
  for (int i = 0; i < 6; ++i) {
    // max over the grid (positive doubles order like their bit patterns)
    atomicMax((unsigned long long *)&err[i], (unsigned long long)__double_as_longlong(e[i]));
  }
}

int main() {
  double *d, h[6] = {0, 0, 0, 0, 0, 0};
  (void)hipMalloc(&d, sizeof h);
  (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, d, 12345ull);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const double ulp = 1.1102230246251565e-16;  // 2^-53
  printf("v_rcp_f64: raw %.3e (2^%.1f)  +1 Newton step %.3e (%.2f x 2^-53)  +2 steps %.3e (%.2f x 2^-53)\n", h[0], log2(h[0]), h[1], h[1] / ulp,
         h[2], h[2] / ulp);
  printf("v_rsq_f64: raw %.3e (2^%.1f)  +1 Newton step %.3e (%.2f x 2^-53)  +2 steps %.3e (%.2f x 2^-53)\n", h[3], log2(h[3]), h[4], h[4] / ulp,
         h[5], h[5] / ulp);
  return 0;
}
