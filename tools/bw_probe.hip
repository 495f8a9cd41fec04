// Read-bandwidth probe for MI355X: streaming double2 reads with a trivial reduction.
//   variant 0: one contiguous stream per workgroup chunk
//   variant 1: 8 row streams (k-major [8][n]) like the engine's weight array
//   variant 2: tile-interleaved [(n/64)][8][64] (a wave reads 8 KB contiguous)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int VAR>
__global__ __launch_bounds__(256) void rd(const double2 *__restrict__ a, size_t npairs, uint32_t chunk, double *out, int rev) {
  const uint32_t begin = blockIdx.x * chunk;
  const uint32_t end = min((size_t)begin + chunk, npairs);
  double acc = 0.0;
  const uint32_t cnt = (end - begin) / 256;
  for (uint32_t t = 0; t < cnt; ++t) {
    const uint32_t i = begin + threadIdx.x + (rev ? (cnt - 1 - t) : t) * 256;
    double2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      size_t idx;
      if (VAR == 0) idx = (size_t)begin * 8 + (size_t)(i - begin) + (size_t)k * (end - begin);  // 8 sub-streams inside the chunk
      else if (VAR == 1) idx = (size_t)k * npairs + i;
      else idx = (((size_t)(i >> 6) * 8 + k) << 6) | (i & 63);
      v[k] = a[idx];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].y;
  }
  if (acc == 123.456) out[0] = acc;
}

// read-modify-write of two k-major [8][n] arrays (the first pass' traffic: R w, R gam, W gam, W w)
template <typename T>
__global__ __launch_bounds__(256) void rw(T *__restrict__ w, T *__restrict__ g, size_t nitems, uint32_t chunk) {
  const uint32_t begin = blockIdx.x * chunk;
  const uint32_t end = min((size_t)begin + chunk, nitems);
  for (uint32_t i = begin + threadIdx.x; i < end; i += 256) {
    T a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a[k] = w[(size_t)k * nitems + i];
      b[k] = g[(size_t)k * nitems + i];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if constexpr (sizeof(T) == 16) {
        g[(size_t)k * nitems + i] = T{b[k].x + a[k].x, b[k].y + a[k].y};
        w[(size_t)k * nitems + i] = T{b[k].x - a[k].x, b[k].y - a[k].y};
      } else {
        g[(size_t)k * nitems + i] = b[k] + a[k];
        w[(size_t)k * nitems + i] = b[k] - a[k];
      }
    }
  }
}

template <typename T>
void probe_rw(const char *name) {
  const size_t n = 1u << 20;  // individuals; 8 rows x 8 B x n = 64 MB per array
  const size_t nitems = n * 8 / sizeof(T);
  T *w, *g;
  CK(hipMalloc(&w, n * 64)); CK(hipMalloc(&g, n * 64));
  CK(hipMemset(w, 0, n * 64)); CK(hipMemset(g, 0, n * 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (uint32_t grid : {256u, 512u, 1024u, 2048u, 4096u}) {
    uint32_t chunk = (nitems + grid - 1) / grid; chunk = (chunk + 255) / 256 * 256;
    const uint32_t gr = (nitems + chunk - 1) / chunk;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(rw<T>, dim3(gr), dim3(256), 0, 0, w, g, nitems, chunk);
    CK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(rw<T>, dim3(gr), dim3(256), 0, 0, w, g, nitems, chunk);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("rw %s (R 128 MB + W 128 MB) grid %4u: %.2f us/launch  %.0f GB/s\n", name, gr, ms / reps * 1e3, 4.0 * n * 64 / (ms / reps * 1e-3) / 1e9);
  }
  // device-to-device copy of the same volume, for reference
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 20; ++r) { CK(hipMemcpyAsync(w, g, n * 64, hipMemcpyDeviceToDevice, 0)); CK(hipMemcpyAsync(g, w, n * 64, hipMemcpyDeviceToDevice, 0)); }
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("hipMemcpy D2D 2 x 64 MB (R 128 MB + W 128 MB): %.2f us  %.0f GB/s\n", ms / 20 * 1e3, 4.0 * n * 64 / (ms / 20 * 1e-3) / 1e9);
  CK(hipFree(w)); CK(hipFree(g));
}

int main(int argc, char **argv) {
  if (argc > 1 && argv[1][0] == 'w') {
    probe_rw<double>("8-byte");
    probe_rw<double2>("16-byte");
    return 0;
  }
  const size_t sizes_mb[] = {64, 256, 2048};
  for (size_t mb : sizes_mb) {
    const size_t bytes = mb << 20;
    const size_t npairs = bytes / 16 / 8;  // pairs per row; 8 rows
    double2 *a; double *out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&out, 8));
    CK(hipMemset(a, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int alt = 0; alt < 2; ++alt)
    for (int var = 1; var < 3; ++var)
      for (uint32_t grid : {256u, 512u, 1024u}) {
        uint32_t chunk = (npairs + grid - 1) / grid; chunk = (chunk + 255) / 256 * 256;
        const uint32_t g = (npairs + chunk - 1) / chunk;
        int flip = 0;
        auto launch = [&]() {
          const int rev = alt ? (flip ^= 1) : 0;
          if (var == 0) hipLaunchKernelGGL(rd<0>, dim3(g), dim3(256), 0, 0, a, npairs, chunk, out, rev);
          else if (var == 1) hipLaunchKernelGGL(rd<1>, dim3(g), dim3(256), 0, 0, a, npairs, chunk, out, rev);
          else hipLaunchKernelGGL(rd<2>, dim3(g), dim3(256), 0, 0, a, npairs, chunk, out, rev);
        };
        for (int w = 0; w < 3; ++w) launch();
        CK(hipEventRecord(e0, 0));
        const int reps = 20;
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("size %4zu MB alt %d var %d grid %4u: %.2f us/launch  %.0f GB/s\n", mb, alt, var, g, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
      }
    CK(hipFree(a)); CK(hipFree(out));
  }
  return 0;
}
