// Go/no-go probe for a resident plain-pass kernel: how long does one in-launch "all workgroups
// contribute a row of 16 doubles, everybody gets the fixed-order total" exchange take on MI355X
// when done as a two-level granule all-reduce (MI355X_MICROARCH.md, price list: R2 granules =
// one aligned 8-byte {tag, 32-bit value} written by one sc1 store; the data is the flag)?
//   level 1: the 32 workgroups of a group (blockIdx % 8) publish their rows; the group's leader sweeps
//            the 32 x 32 granules until every tag matches, adds the rows in workgroup order, publishes;
//   level 2: every workgroup sweeps the 8 x 32 leader granules, adds the 8 rows in group order.
// Every spin is bounded (2 ms) and a failure sets an abort word that ends all later waits.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/xchg_probe tools/xchg_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef unsigned long long u64;
constexpr int kGroups = 8, kPerGroup = 32, kJ = 16, kGran = 2 * kJ;  // 32 granules per row

struct Xb {
  u64 rows[kGroups][kPerGroup][kGran];  // level 1
  u64 sums[2][kGroups][kGran];          // level 2, double-buffered by pass parity
  u64 abort_word;
};

__device__ __forceinline__ void put(u64 *g, unsigned tag, unsigned v) {
  __hip_atomic_store(g, ((u64)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 get(const u64 *g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one wave: re-read N granules per lane (index lane + 64 i) until every tag == tag
template <int N>
__device__ __forceinline__ bool sweep(const u64 *base, unsigned tag, unsigned (&v)[N], u64 *abort_word) {
  const unsigned lane = threadIdx.x & 63u;
  const u64 t0 = wall_clock64();
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const u64 x = get(base + lane + 64 * i);
      v[i] = (unsigned)x;
      ok &= (unsigned)(x >> 32) == tag;
    }
    if (__all(ok)) return true;
    if (wall_clock64() - t0 > 200000ull || get(abort_word) != 0ull) {  // 2 ms
      if (lane == 0) __hip_atomic_store(abort_word, (u64)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

__device__ __forceinline__ double join(unsigned lo, unsigned hi) { return __longlong_as_double(((u64)hi << 32) | lo); }

// work_ns: stand-in for the register-resident sweep between two exchanges
__global__ __launch_bounds__(512) void probe(Xb *xb, int passes, unsigned tag0, unsigned work_ticks, double *out) {
  __shared__ double s_tot[kJ];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned g = blockIdx.x % kGroups, m = blockIdx.x / kGroups;  // group, member
  double acc = (double)(blockIdx.x + 1);
  bool alive = true;
  for (int p = 0; p < passes && alive; ++p) {
    const unsigned tag = tag0 + (unsigned)p + 1u;
    if (work_ticks) {
      const u64 t0 = wall_clock64();
      while (wall_clock64() - t0 < work_ticks) __builtin_amdgcn_s_sleep(2);
    }
    // the workgroup's row: value j = acc * (j + 1)  (thread j < 16 owns value j)
    if (tid < kJ) {
      const u64 bits = __double_as_longlong(acc * (double)(tid + 1));
      put(&xb->rows[g][m][2 * tid], tag, (unsigned)bits);
      put(&xb->rows[g][m][2 * tid + 1], tag, (unsigned)(bits >> 32));
    }
    // level 1: leader = member 0 of the group, wave 0
    if (m == 0 && wave == 0) {
      unsigned v[16];  // lane l: granule c = l % 32 of members 2 i + (l >= 32)
      alive = sweep<16>(&xb->rows[g][0][0], tag, v, &xb->abort_word);
      // lo/hi halves sit in neighbouring lanes (c even = lo, c odd = hi): rebuild the doubles in even lanes
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const unsigned other = __shfl_xor((int)v[i], 1);
        const double d = (lane & 1u) ? join(other, v[i]) : join(v[i], other);
        s += d;  // members 2 i + (lane >= 32), ascending
      }
      s += __shfl_xor(s, 32);  // even members + odd members
      if (lane < 32u && !(lane & 1u)) {  // lane 2 j holds value j
        const u64 bits = __double_as_longlong(s);
        put(&xb->sums[p & 1][g][lane], tag, (unsigned)bits);
        put(&xb->sums[p & 1][g][lane + 1], tag, (unsigned)(bits >> 32));
      }
    }
    // level 2: wave 0 of every workgroup
    if (wave == 0) {
      unsigned v[4];  // lane l: granule c = l % 32 of groups 2 i + (l >= 32)
      alive = sweep<4>(&xb->sums[p & 1][0][0], tag, v, &xb->abort_word);
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned other = __shfl_xor((int)v[i], 1);
        s += (lane & 1u) ? join(other, v[i]) : join(v[i], other);
      }
      s += __shfl_xor(s, 32);
      if (lane < 32u && !(lane & 1u)) s_tot[lane >> 1] = s;
    }
    __syncthreads();
    alive = __syncthreads_and(alive ? 1 : 0) != 0;
    acc = acc * 0.5 + s_tot[0] * 1e-6;  // the next pass depends on the total
  }
  if (tid == 0) out[blockIdx.x] = alive ? acc : -1.0;
}

int main() {
  Xb *xb;
  double *out;
  const int grid = kGroups * kPerGroup;
  CK(hipMalloc(&xb, sizeof(Xb)));
  CK(hipMemset(xb, 0, sizeof(Xb)));
  CK(hipMalloc(&out, grid * sizeof(double)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  unsigned tag0 = 0;
  for (unsigned work_ns : {0u, 1000u, 2000u}) {
    for (int passes : {9, 900}) {
      hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 0, 0, xb, passes, tag0, work_ns / 10u, out);  // warm-up
      tag0 += passes;
      CK(hipDeviceSynchronize());
      const int reps = passes == 9 ? 50 : 3;
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 0, 0, xb, passes, tag0, work_ns / 10u, out);
        tag0 += passes;
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<double> h(grid);
      CK(hipMemcpy(h.data(), out, grid * sizeof(double), hipMemcpyDeviceToHost));
      int bad = 0;
      for (double v : h) bad += v < 0 || v != h[0] - (h[0] - v);  // (all finite)
      for (double v : h) bad += v < 0;
      printf("work %4u ns, %3d passes per launch: %.2f us per launch, %.2f us per pass (exchange + work), aborted workgroups %d\n",
             work_ns, passes, ms / reps * 1e3, ms / reps * 1e3 / passes, bad);
    }
  }
  return 0;
}
