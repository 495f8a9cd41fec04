// Go/no-go probe for a resident plain-pass kernel: how long does one in-launch "all workgroups
// contribute a row of 16 doubles, everybody gets the fixed-order total" exchange take on MI355X
// when done as a two-level granule all-reduce (MI355X_MICROARCH.md, price list: R2 granules =
// one aligned 8-byte {tag, 32-bit value} written by one sc1 store; the data is the flag)?
//   level 1: the 32 workgroups of a group (blockIdx % 8) publish their rows; the group's leader sweeps
//            the 32 x 32 granules until every tag matches, adds the rows in workgroup order, publishes;
//   level 2: every workgroup sweeps the 8 x 32 leader granules, adds the 8 rows in group order.
// Every spin is bounded (2 ms) and a failure sets an abort word that ends all later waits.
// Variants (second table): the workgroups of a group are the workgroups of one XCD (round-robin dispatch:
// blockIdx % 8 == XCC_ID, checked and reported), so level 1 can use workgroup-scope accesses (sc0: bypass
// the CU's L1, served by the XCD's L2) instead of agent-scope ones (sc1: coherent across the 8 L2s);
// variant 2 also hands the total back through the L2 (leaders exchange among themselves at agent scope).
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
  u64 bc[2][kGroups][kGran];            // variant 2: the total, handed back inside the XCD
  u64 rep[2][32][512];                  // variants 7+: replicas of sums (4 KB apart: 2 KB of granules + padding), so that the
                                        // 256 pollers do not all read the same 2 KB through one memory channel
  u64 flat[2][kGroups * kPerGroup][kGran];  // variant 25: one level -- every workgroup polls every workgroup's row
  u64 abort_word;
  u64 xcc_mismatch;                     // workgroups whose XCC_ID is not blockIdx % 8
};

template <bool L2 = false>
__device__ __forceinline__ void put(u64 *g, unsigned tag, unsigned v) {
  if (L2)
    __hip_atomic_store(g, ((u64)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    __hip_atomic_store(g, ((u64)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 get(const u64 *g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// L2 poll: N atomic ORs of 0 in flight (they execute at the XCD's L2; a workgroup-scope LOAD may be served by the
// CU's own L1 and never see another CU's store, and the compiler turns an idempotent relaxed RMW back into a load)
// -DL2MODE=1 / 2: instead, invalidate the CU's L1 (buffer_inv sc0 / sc1) and poll with plain loads; 3: plain loads with nt
#ifndef L2MODE
#define L2MODE 0
#endif
__device__ __forceinline__ void l2_issue(u64 &x, const u64 *addr) {
#if L2MODE == 0
  const u64 zero = 0ull;
  asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0" : "=&v"(x) : "v"(addr), "v"(zero) : "memory");
#elif L2MODE == 3
  asm volatile("global_load_dwordx2 %0, %1, off nt" : "=&v"(x) : "v"(addr) : "memory");
#else
  asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(x) : "v"(addr) : "memory");
#endif
}
template <int N>
__device__ __forceinline__ void l2_poll(const u64 *base, unsigned lane, u64 (&x)[N]) {
#if L2MODE == 1
  asm volatile("buffer_inv sc0" ::: "memory");
#elif L2MODE == 2
  asm volatile("buffer_inv sc1" ::: "memory");
#endif
#pragma unroll
  for (int i = 0; i < N; ++i) l2_issue(x[i], base + lane + 64 * i);
  if constexpr (N == 16)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                 "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) : : "memory");
  else if constexpr (N == 4)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : : "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]) : : "memory");
}

// diagnostics (-DPOLLSTATS): polls and 10 ns ticks spent in the sweeps of level 1 (index 0) and level 2 (index 1), summed
// over a launch by workgroup 0 (a leader) and workgroup 9 (a member)
__device__ unsigned long long g_polls[2][2], g_ticks[2][2];
// one wave: re-read N granules per lane (index lane + 64 i) until every tag == tag
// done (variant 3): several waves of the workgroup poll the same granules out of phase (stagger ticks apart);
// the first one to see them complete posts the tag in LDS and the others leave (return value 2)
template <int N, bool L2 = false>
__device__ __forceinline__ int sweep(const u64 *base, unsigned tag, unsigned (&v)[N], u64 *abort_word,
                                     volatile unsigned *done = nullptr, unsigned stagger = 0u) {
  const unsigned lane = threadIdx.x & 63u;
  const u64 t0 = wall_clock64();
  if (stagger) {
    while (wall_clock64() - t0 < stagger) __builtin_amdgcn_s_sleep(1);
  }
  for (;;) {
    if (done && *done == tag) return 2;
    bool ok = true;
    u64 x[N];
    if constexpr (L2) {
      l2_poll<N>(base, lane, x);
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) x[i] = get(base + lane + 64 * i);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      v[i] = (unsigned)x[i];
      ok &= (unsigned)(x[i] >> 32) == tag;
    }
#ifdef POLLSTATS
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 9)) atomicAdd(&g_polls[blockIdx.x == 9][N == 4], 1ull);
#endif
    if (__all(ok)) {
#ifdef POLLSTATS
      if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 9)) atomicAdd(&g_ticks[blockIdx.x == 9][N == 4], wall_clock64() - t0);
#endif
      if (done) *done = tag;
      return 1;
    }
    if (wall_clock64() - t0 > 200000ull || get(abort_word) != 0ull) {  // 2 ms
      if (lane == 0) __hip_atomic_store(abort_word, (u64)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return 0;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

__device__ __forceinline__ double join(unsigned lo, unsigned hi) { return __longlong_as_double(((u64)hi << 32) | lo); }

// work_ns: stand-in for the register-resident sweep between two exchanges
template <int VAR>
__global__ __launch_bounds__(512) void probe(Xb *xb, int passes, unsigned tag0, unsigned work_ticks, double *out, unsigned stagger_ticks) {
  __shared__ double s_tot[kJ];
  __shared__ unsigned s_done[2];
  constexpr bool F1 = VAR == 1 || VAR == 2;
  constexpr unsigned kPollers = VAR == 3 ? 4u : 1u;  // variant 3: four waves poll out of phase
  if (threadIdx.x < 2) s_done[threadIdx.x] = 0u;
  __syncthreads();
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned g = blockIdx.x % kGroups, m = blockIdx.x / kGroups;  // group, member
  {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (tid == 0 && (xcc & 15u) != g) atomicAdd(&xb->xcc_mismatch, 1ull);
  }
  double acc = (double)(blockIdx.x + 1);
  bool alive = true;
  for (int p = 0; p < passes && alive; ++p) {
    const unsigned tag = tag0 + (unsigned)p + 1u;
    if (work_ticks) {
      const u64 t0 = wall_clock64();
      while (wall_clock64() - t0 < work_ticks) __builtin_amdgcn_s_sleep(2);
    }
    if constexpr (VAR == 25) {
      // ONE level: every workgroup stores its row, waits `stagger_ticks` (the expected skew), and its waves 0..3 each
      // poll 64 of the 256 rows (32 loads per lane); partial sums through LDS, added in wave order
      __shared__ double s_part[4][kJ];
      if (tid < kJ) {
        const u64 bits = __double_as_longlong(acc * (double)(tid + 1));
        put(&xb->flat[p & 1][blockIdx.x][2 * tid], tag, (unsigned)bits);
        put(&xb->flat[p & 1][blockIdx.x][2 * tid + 1], tag, (unsigned)(bits >> 32));
      }
      if (wave < 4) {
        unsigned v[32];
        alive = sweep<32>(&xb->flat[p & 1][64 * wave][0], tag, v, &xb->abort_word, nullptr, stagger_ticks) != 0;
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const unsigned other = __shfl_xor((int)v[i], 1);
          s += (lane & 1u) ? join(other, v[i]) : join(v[i], other);
        }
        s += __shfl_xor(s, 32);
        if (lane < 32u && !(lane & 1u)) s_part[wave][lane >> 1] = s;
      }
      __syncthreads();
      if (tid < kJ) s_tot[tid] = ((s_part[0][tid] + s_part[1][tid]) + s_part[2][tid]) + s_part[3][tid];
      __syncthreads();
      alive = __syncthreads_and(alive ? 1 : 0) != 0;
      acc = acc * 0.5 + s_tot[0] * 1e-6;
      continue;
    }
    // the workgroup's row: value j = acc * (j + 1)  (thread j < 16 owns value j)
    if (tid < kJ) {
      const u64 bits = __double_as_longlong(acc * (double)(tid + 1));
      put<F1>(&xb->rows[g][m][2 * tid], tag, (unsigned)bits);
      put<F1>(&xb->rows[g][m][2 * tid + 1], tag, (unsigned)(bits >> 32));
    }
    // level 1: leader = member 0 of the group, wave 0
    int r1 = 1;
    if (m == 0 && wave < kPollers) {
      unsigned v[16];  // lane l: granule c = l % 32 of members 2 i + (l >= 32)
      r1 = sweep<16, F1>(&xb->rows[g][0][0], tag, v, &xb->abort_word, VAR == 3 ? &s_done[0] : nullptr, VAR == 3 ? wave * stagger_ticks : 0u);
      alive = r1 != 0;
      // lo/hi halves sit in neighbouring lanes (c even = lo, c odd = hi): rebuild the doubles in even lanes
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const unsigned other = __shfl_xor((int)v[i], 1);
        const double d = (lane & 1u) ? join(other, v[i]) : join(v[i], other);
        s += d;  // members 2 i + (lane >= 32), ascending
      }
      s += __shfl_xor(s, 32);  // even members + odd members
      if (r1 == 1 && lane < 32u && !(lane & 1u)) {  // lane 2 j holds value j
        const u64 bits = __double_as_longlong(s);
        if (VAR == 7) {
          for (unsigned r = 0; r < stagger_ticks; ++r) {  // (stagger_ticks = number of replicas here)
            put(&xb->rep[p & 1][r][g * kGran + lane], tag, (unsigned)bits);
            put(&xb->rep[p & 1][r][g * kGran + lane + 1], tag, (unsigned)(bits >> 32));
          }
        } else {
          put(&xb->sums[p & 1][g][lane], tag, (unsigned)bits);
          put(&xb->sums[p & 1][g][lane + 1], tag, (unsigned)(bits >> 32));
        }
      }
    }
    // level 2: wave 0 of every workgroup (variant 2: of the leaders only, which then hand the total back through the L2)
    if (wave < kPollers && (VAR != 2 || m == 0)) {
      unsigned v[4];  // lane l: granule c = l % 32 of groups 2 i + (l >= 32)
      const u64 *src = VAR == 7 ? &xb->rep[p & 1][blockIdx.x % stagger_ticks][0] : &xb->sums[p & 1][0][0];
      // VAR == 9: nobody polls level 2 before it can be there: members wait (stagger_ticks & 0xffff) ticks after their own
      // store, leaders (stagger_ticks >> 16) ticks after theirs -- fewer useless polls hammering the 2 KB of group sums
      const unsigned quiet = VAR == 9 ? (m != 0 ? (stagger_ticks & 0xffffu) : (stagger_ticks >> 16)) : VAR == 3 ? wave * stagger_ticks : 0u;
      const int r2 = sweep<4>(src, tag, v, &xb->abort_word, VAR == 3 ? &s_done[1] : nullptr, quiet);
      alive = r2 != 0 && alive;
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned other = __shfl_xor((int)v[i], 1);
        s += (lane & 1u) ? join(other, v[i]) : join(v[i], other);
      }
      s += __shfl_xor(s, 32);
      if (VAR == 2) {
        if (lane < 32u && !(lane & 1u)) {
          const u64 bits = __double_as_longlong(s);
          put<true>(&xb->bc[p & 1][g][lane], tag, (unsigned)bits);
          put<true>(&xb->bc[p & 1][g][lane + 1], tag, (unsigned)(bits >> 32));
        }
      }
      if (r2 == 1 && lane < 32u && !(lane & 1u)) s_tot[lane >> 1] = s;
    }
    if (VAR == 2 && wave == 0 && m != 0) {
      unsigned v[1];  // lanes 0..31: the 32 granules of the group's total (lanes 32..63 re-read them)
      alive = sweep<1, true>(&xb->bc[p & 1][g][0] - (lane & 32u), tag, v, &xb->abort_word);
      const unsigned other = __shfl_xor((int)v[0], 1);
      const double s = (lane & 1u) ? join(other, v[0]) : join(v[0], other);
      if (lane < 32u && !(lane & 1u)) s_tot[lane >> 1] = s;
    }
    __syncthreads();
    alive = __syncthreads_and(alive ? 1 : 0) != 0;
    acc = acc * 0.5 + s_tot[0] * 1e-6;  // the next pass depends on the total
  }
  if (tid == 0) {
    out[blockIdx.x] = alive ? acc : -1.0;
    out[gridDim.x + blockIdx.x] = s_tot[1];  // the last total, as this workgroup saw it
  }
}

// variants 9 ..: (leader quiet ticks << 16) | member quiet ticks, 10 ns each
static const unsigned g_quiet[] = {100u, 140u, 170u, 200u, (30u << 16) | 140u, (30u << 16) | 170u, (50u << 16) | 170u, (50u << 16) | 200u};
int main(int argc, char **argv) {
  Xb *xb;
  double *out;
  const int grid = kGroups * kPerGroup;
  // argv[1]: 0 = hipMalloc (default), 1 = fine-grained, 2 = uncached device memory for the exchange buffer
  const int kind = argc > 1 ? atoi(argv[1]) : 0;
  const int max_var = argc > 2 ? atoi(argv[2]) : 9;
  const int min_var = argc > 3 ? atoi(argv[3]) : 0;
  if (kind == 0) CK(hipMalloc(&xb, sizeof(Xb)));
  if (kind == 1) CK(hipExtMallocWithFlags((void **)&xb, sizeof(Xb), hipDeviceMallocFinegrained));
  if (kind == 2) CK(hipExtMallocWithFlags((void **)&xb, sizeof(Xb), hipDeviceMallocUncached));
  printf("exchange buffer: %s\n", kind == 0 ? "hipMalloc" : kind == 1 ? "fine-grained" : "uncached");
  CK(hipMemset(xb, 0, sizeof(Xb)));
  CK(hipMalloc(&out, 2 * grid * sizeof(double)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  unsigned tag0 = 0;
  auto launch = [&](int var, int passes, unsigned ticks) {
    if (var == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 0, 0, xb, passes, tag0, ticks, out, 0u);
    if (var == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(512), 0, 0, xb, passes, tag0, ticks, out, 0u);
    if (var == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(512), 0, 0, xb, passes, tag0, ticks, out, 0u);
    if (var == 3) hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(512), 0, 0, xb, passes, tag0, ticks, out, 15u);
    if (var >= 4 && var < 9) hipLaunchKernelGGL(probe<7>, dim3(grid), dim3(512), 0, 0, xb, passes, tag0, ticks, out, 1u << (var - 3));
    if (var >= 9 && var < 17) hipLaunchKernelGGL(probe<9>, dim3(grid), dim3(512), 0, 0, xb, passes, tag0, ticks, out, g_quiet[var - 9]);
    if (var >= 17) hipLaunchKernelGGL(probe<25>, dim3(grid), dim3(512), 0, 0, xb, passes, tag0, ticks, out, (unsigned)(var - 17) * 20u);
    tag0 += passes;
  };
  for (int var = min_var; var < max_var; ++var) {
    printf("variant %d (%s)\n", var, var == 0 ? "agent scope on both levels" : var == 1 ? "level 1 through the XCD's L2 (sc0)" :
           var == 2 ? "level 1 through the L2, leaders all-to-all at agent scope, total handed back through the L2" :
           var == 3 ? "agent scope, four waves poll out of phase (150 ns apart)" :
           var < 9 ? "agent scope, the group sums replicated 2 / 4 / 8 / 16 / 32 times (variants 4 .. 8), 4 KB apart; a workgroup polls replica blockIdx % R" :
           var < 17 ? "agent scope; no level-2 poll before it can succeed: members / leaders stay quiet 1.0/0, 1.4/0, 1.7/0, 2.0/0, 1.4/0.3, 1.7/0.3, 1.7/0.5, 2.0/0.5 us (variants 9 .. 16)" :
           "ONE level: every workgroup polls all 256 rows (64 KB) with four waves, first poll 0 / 0.2 / 0.4 / 0.6 / 0.8 us after its own store (variants 17 .. 21)");
    for (unsigned work_ns : {0u, 1000u, 2000u}) {
      for (int passes : {9, 900}) {
        launch(var, passes, work_ns / 10u);  // warm-up
        CK(hipDeviceSynchronize());
        const int reps = passes == 9 ? 50 : 3;
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) launch(var, passes, work_ns / 10u);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<double> h(2 * grid);
        CK(hipMemcpy(h.data(), out, 2 * grid * sizeof(double), hipMemcpyDeviceToHost));
        int bad = 0, differ = 0;
        for (int i = 0; i < grid; ++i) bad += h[i] < 0;
        for (int i = 0; i < grid; ++i) differ += h[grid + i] != h[grid];  // every workgroup must have seen the same total
        u64 mism = 0;
        CK(hipMemcpy(&mism, &xb->xcc_mismatch, sizeof mism, hipMemcpyDeviceToHost));
#ifdef POLLSTATS
        {
          unsigned long long hp[2][2], ht[2][2], z[2][2] = {{0, 0}, {0, 0}};
          CK(hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_polls), sizeof hp));
          CK(hipMemcpyFromSymbol(ht, HIP_SYMBOL(g_ticks), sizeof ht));
          CK(hipMemcpyToSymbol(HIP_SYMBOL(g_polls), z, sizeof z));
          CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ticks), z, sizeof z));
          const double ex = (double)passes * (reps + 1);
          printf("    per exchange: leader level 1: %.2f polls, %.2f us; leader level 2: %.2f polls, %.2f us; member level 2: %.2f polls, %.2f us\n",
                 hp[0][0] / ex, ht[0][0] * 0.01 / ex, hp[0][1] / ex, ht[0][1] * 0.01 / ex, hp[1][1] / ex, ht[1][1] * 0.01 / ex);
        }
#endif
        printf("  work %4u ns, %3d passes per launch: %.2f us per launch, %.2f us per pass (exchange + work), aborted workgroups %d, "
               "workgroups with a different total %d, XCC_ID != blockIdx %% 8 so far: %llu\n", work_ns, passes, ms / reps * 1e3,
               ms / reps * 1e3 / passes, bad, differ, mism);
      }
    }
  }
  return 0;
}
