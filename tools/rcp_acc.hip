// Accuracy of v_rcp_f64 / v_rsq_f64 with 0, 1, 2 Newton steps on gfx950 (why fast_rcp / fast_rsqrt take two: 24 bits raw,
// 20-37 ulp after one step, 1.0-1.2 ulp after two).   build: hipcc --offload-arch=gfx950 -O3 -o tools/rcp_acc tools/rcp_acc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double *x, double *o0, double *o1, double *o2, double *q0, double *q1, double *q2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  o0[i] = r;
  double e = fma(-v, r, 1.0);
  r = fma(r, e, r);
  o1[i] = r;
  e = fma(-v, r, 1.0);
  r = fma(r, e, r);
  o2[i] = r;
  double y = __builtin_amdgcn_rsq(v);
  q0[i] = y;
  double f = fma(-(v * y), y, 1.0);
  y = fma(0.5 * y, f, y);
  q1[i] = y;
  f = fma(-(v * y), y, 1.0);
  y = fma(0.5 * y, f, y);
  q2[i] = y;
}
int main() {
  const int n = 1 << 22;
  std::vector<double> h(n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> u(-30.0, 30.0), m(1.0, 2.0);
  for (int i = 0; i < n; ++i) h[i] = std::ldexp(m(g), (int)u(g));
  double *d[7];
  for (auto &p : d) hipMalloc(&p, n * sizeof(double));
  hipMemcpy(d[0], h.data(), n * sizeof(double), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], d[6], n);
  hipDeviceSynchronize();
  const char *names[6] = {"rcp raw", "rcp + 1 Newton", "rcp + 2 Newton", "rsq raw", "rsq + 1 Newton", "rsq + 2 Newton"};
  for (int t = 0; t < 6; ++t) {
    std::vector<double> o(n);
    hipMemcpy(o.data(), d[t + 1], n * sizeof(double), hipMemcpyDeviceToHost);
    long double worst = 0;
    for (int i = 0; i < n; ++i) {
      long double exact = t < 3 ? 1.0L / (long double)h[i] : 1.0L / sqrtl((long double)h[i]);
      long double rel = fabsl(((long double)o[i] - exact) / exact);
      if (rel > worst) worst = rel;
    }
    printf("%-16s max relative error %.3Le (%.2Lf ulp of 2^-53)\n", names[t], worst, worst / 1.1102230246251565e-16L);
  }
  return 0;
}
