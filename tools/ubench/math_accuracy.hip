// Accuracy of the device formulation's building blocks (csrc/tsamd_device.h) against long-double references on the host:
//   exp_digamma_split -> z * exp(a) = exp(psi(x)),  exp_nonpos,  fast_rcp,  fast_rsqrt.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../terastructure_amd/csrc -o math_accuracy math_accuracy.hip
#include "tsamd_device.h"
#include <cmath>
#include <cstdio>
#include <vector>

using namespace tsamd;

__global__ void eval(const double *x, double *z, double *a, double *ex, double *rc, double *rs, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  exp_digamma_split(x[i], z[i], a[i]);
  ex[i] = exp_nonpos(-x[i] * 1e-3 * 40.0 / 1e4);  // arguments in [-40, 0] for x in [1e-3, 1e7] (see the host)
  rc[i] = fast_rcp(x[i]);
  rs[i] = fast_rsqrt(x[i]);
}

static long double psi_l(long double x) {
  long double s = 0.0L;
  while (x < 30.0L) {
    s -= 1.0L / x;
    x += 1.0L;
  }
  const long double f = 1.0L / (x * x);
  const long double t = f * (-1.0L / 12 + f * (1.0L / 120 + f * (-1.0L / 252 + f * (1.0L / 240 + f * (-1.0L / 132 + f * (691.0L / 32760 + f * (-1.0L / 12)))))));
  return s + logl(x) - 0.5L / x + t;
}

int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), z(n), a(n), ex(n), rc(n), rs(n);
  unsigned long long s = 99;
  for (int i = 0; i < n; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const double u = (double)(s >> 11) * (1.0 / 9007199254740992.0);
    x[i] = pow(10.0, -3.0 + 10.0 * u);  // [1e-3, 1e7], log-uniform
  }
  double *dx, *dz, *da, *de, *dc, *ds;
  const size_t b = n * sizeof(double);
  (void)hipMalloc(&dx, b); (void)hipMalloc(&dz, b); (void)hipMalloc(&da, b); (void)hipMalloc(&de, b); (void)hipMalloc(&dc, b); (void)hipMalloc(&ds, b);
  (void)hipMemcpy(dx, x.data(), b, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(eval, dim3(n / 256), dim3(256), 0, 0, dx, dz, da, de, dc, ds, n);
  (void)hipMemcpy(z.data(), dz, b, hipMemcpyDeviceToHost); (void)hipMemcpy(a.data(), da, b, hipMemcpyDeviceToHost);
  (void)hipMemcpy(ex.data(), de, b, hipMemcpyDeviceToHost); (void)hipMemcpy(rc.data(), dc, b, hipMemcpyDeviceToHost);
  (void)hipMemcpy(rs.data(), ds, b, hipMemcpyDeviceToHost);
  long double e_psi = 0, e_a = 0, e_exp = 0, e_rc = 0, e_rs = 0;
  for (int i = 0; i < n; ++i) {
    const long double xl = x[i];
    const long double want_a = psi_l(xl) - logl(xl + 10.0L);   // a = psi(x) - log(z)
    e_a = fmaxl(e_a, fabsl((long double)a[i] - want_a));        // absolute: a is an exponent
    const long double got = (long double)z[i] * expl((long double)a[i]), want = expl(psi_l(xl));
    e_psi = fmaxl(e_psi, fabsl(got - want) / want);
    const double d = -x[i] * 1e-3 * 40.0 / 1e4;
    if (d >= -700.0) e_exp = fmaxl(e_exp, fabsl((long double)ex[i] - expl((long double)d)) / expl((long double)d));
    e_rc = fmaxl(e_rc, fabsl((long double)rc[i] * xl - 1.0L));
    e_rs = fmaxl(e_rs, fabsl((long double)rs[i] * sqrtl(xl) - 1.0L));
  }
  printf("exp_digamma_split: max |a - (psi(x) - log z)| = %.3Le   max rel error of z exp(a) = %.3Le\n", e_a, e_psi);
  printf("exp_nonpos on [-40, 0]: max rel error %.3Le   fast_rcp: %.3Le   fast_rsqrt: %.3Le   (2^-53 = 1.11e-16)\n", e_exp, e_rc, e_rs);
  return 0;
}
