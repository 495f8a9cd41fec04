// Go / no-go probe for overlapping consecutive pass launches (MI355X):
// a chain of dependent "passes" (each: wait for the previous pass' per-workgroup flags, stream
// 64 MB, raise own flags) launched
//   mode 0: on one stream, dependency = stream order (what the engine does today)
//   mode 1: alternately on two streams, dependency = the flags only, so that the launch /
//           dispatch / first-load latency of pass i+1 overlaps the sweep of pass i
// and in both cases eagerly or as a captured graph.  Prints microseconds per pass.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int kBlock = 512;
constexpr int kK = 8;

struct Flags { unsigned long long seq[2][256]; double rows[2][256][16]; };

__global__ __launch_bounds__(kBlock) void pass(const double2 *__restrict__ w, size_t npairs, uint32_t chunk, Flags *f,
                                               unsigned long long epoch, int wait_flags, int early_prefetch, double *sink) {
  const uint32_t tid = threadIdx.x, begin = blockIdx.x * chunk;
  const uint32_t end = min((size_t)begin + chunk, npairs);
  const uint32_t slot = (uint32_t)(epoch & 1ull), prev = slot ^ 1u;
  double2 buf[kK];
  uint32_t i = begin + tid;
  if (early_prefetch && i < end)
    for (int k = 0; k < kK; ++k) buf[k] = w[(size_t)k * npairs + i];
  __builtin_amdgcn_sched_barrier(0);
  double b = 1.0;
  if (wait_flags) {  // every thread watches one of the previous pass' flags
    if (tid < gridDim.x) {
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(&f->seq[prev][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch - 1ull) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 2000000ull) { if (tid == 0 && blockIdx.x == 0) sink[1] += 1.0; break; }  // 20 ms: count it
      }
    }
    __syncthreads();
  }
  // "rows" of the previous pass -> the scalar this pass depends on
  if (tid < 16) b = __hip_atomic_load(&f->rows[prev][tid % gridDim.x][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  b = __shfl(b, 0) * 1e-300 + 1.0;
  if (!early_prefetch && i < end)
    for (int k = 0; k < kK; ++k) buf[k] = w[(size_t)k * npairs + i];
  double acc = 0.0;
  for (; i < end; i += kBlock) {
    double2 nxt[kK];
    const uint32_t in = (i + kBlock < end) ? i + kBlock : i;
    for (int k = 0; k < kK; ++k) nxt[k] = w[(size_t)k * npairs + in];
    for (int k = 0; k < kK; ++k) acc = fma(buf[k].x + buf[k].y, b, acc);
    for (int k = 0; k < kK; ++k) buf[k] = nxt[k];
  }
  // workgroup "row" + flag
  __shared__ double red[kBlock / 64];
  for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid < 16) {
    double v = 0;
    for (int q = 0; q < kBlock / 64; ++q) v += red[q];
    __hip_atomic_store(&f->rows[slot][blockIdx.x][tid], v * 1e-300, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store(&f->seq[slot][blockIdx.x], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (acc == 123.456) sink[0] = acc;
  }
}

int main(int argc, char **argv) {
  const size_t n = argc > 1 ? atol(argv[1]) : 1000448;  // individuals
  const size_t npairs = n / 2;
  double2 *w; double *sink; Flags *f;
  CK(hipMalloc(&w, npairs * kK * sizeof(double2)));
  CK(hipMemset(w, 0, npairs * kK * sizeof(double2)));
  CK(hipMalloc(&sink, 16)); CK(hipMemset(sink, 0, 16));
  CK(hipMalloc(&f, sizeof(Flags)));
  uint32_t chunk = (npairs + 255) / 256; chunk = (chunk + kBlock - 1) / kBlock * kBlock;
  const uint32_t grid = (npairs + chunk - 1) / chunk;
  hipStream_t s[2];
  CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
  hipEvent_t e0, e1, fork, join;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  const int passes = 160, reps = argc > 2 ? atoi(argv[2]) : 10;
  printf("n = %zu, grid = %u x %d threads, %.1f MB per pass\n", n, grid, kBlock, npairs * kK * 16.0 / 1e6);
  fflush(stdout);
  for (int mode = 0; mode < 2; ++mode)
    for (int graph = 0; graph < 2; ++graph)
      for (int early = 0; early < 2; ++early) {
        CK(hipMemset(f, 0, sizeof(Flags)));
        CK(hipDeviceSynchronize());
        unsigned long long epoch = 1;
        hipGraphExec_t ge = nullptr;
        auto enqueue = [&](unsigned long long base) {
          if (mode == 1) { CK(hipEventRecord(fork, s[0])); CK(hipStreamWaitEvent(s[1], fork, 0)); }
          for (int j = 0; j < passes; ++j) {
            hipStream_t st = s[mode ? (j & 1) : 0];
            hipLaunchKernelGGL(pass, dim3(grid), dim3(kBlock), 0, st, w, npairs, chunk, f, base + j, (j > 0 || base > 1) ? 1 : 0,
                               early, sink);
          }
          if (mode == 1) { CK(hipEventRecord(join, s[1])); CK(hipStreamWaitEvent(s[0], join, 0)); }
        };
        if (graph) {
          // (frozen epochs: the flags are cleared before every replay instead)
          hipGraph_t g;
          CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
          CK(hipMemsetAsync(f, 0, sizeof(Flags), s[0]));
          enqueue(1);
          CK(hipStreamEndCapture(s[0], &g));
          CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        auto run = [&]() {
          if (graph) CK(hipGraphLaunch(ge, s[0]));
          else { enqueue(epoch); epoch += passes; }
        };
        run();
        CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1]));
        CK(hipEventRecord(e0, s[0]));
        for (int r = 0; r < reps; ++r) run();
        CK(hipEventRecord(e1, s[0]));
        CK(hipEventSynchronize(e1));
        CK(hipStreamSynchronize(s[1]));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double to[2] = {0, 0};
        CK(hipMemcpy(to, sink, 16, hipMemcpyDeviceToHost));
        CK(hipMemset(sink, 0, 16));
        printf("%s %s prefetch %s flags: %.2f us per pass (%g flag-wait timeouts)\n", mode ? "two streams" : "one stream ",
               graph ? "graph" : "eager", early ? "before" : "after ", ms * 1e3 / (reps * passes), to[1]);
        fflush(stdout);
      }
  return 0;
}
