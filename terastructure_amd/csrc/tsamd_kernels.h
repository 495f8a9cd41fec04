// HIP kernels of the SNP-minibatch SVI engine (gfx950).  See tsamd_device.h for
// the formulation / state machine and DESIGN.md for layouts and rooflines.
//
// Kernels (all stream-ordered, no host round trip, no inter-workgroup hand-off):
//   ts_pass<K, FIRST, BLOCK, VEC>(p, parity)
//        prologue (every workgroup, redundantly): add up the partial rows of the previous
//        pass in a fixed order and run its K x 2 epilogue (update_lambda, estimate_beta,
//        convergence test); then one inner pass over the shard's individuals for the
//        current SNP: phi for both parents + lambda_t accumulation -> one partial row per
//        workgroup.  FIRST starts the next SNP of the schedule and applies the previous
//        SNP's gamma/Elogtheta step in the same sweep.
//   ts_flush(p, parity)       completes the pending pass (end of a schedule)
//   ts_begin(ctl, n, parity)  starts a schedule
//   ts_reduce_rows(p, parity) sharded: row sum -> ctl->lt for the all-reduce
//   ts_refresh_w<K>           w = exp(psi(gamma) - rowmax) after tsamd_set_gamma
#pragma once
#include "tsamd_device.h"

namespace tsamd {

// ---------------------------------------------------------------------------
// Finish the pending pass described by S: lambda_t[j] = eb_used[j] * sum_rows, then
// update_lambda + estimate_beta + convergence test (src/snpsamplinge.cc:356-364,
// :267-296; abs_mean src/matrix.hh:885-893).  Called by ALL threads of a workgroup.
// Fixed summation order: thread (r, j) adds rows r, r+R, ... (R = BLOCK / J), then
// r = 0..R-1.  Outputs in LDS: s_lam = new lambda[loc], s_eb = new exp(Elogbeta[loc]).
// Returns (uniformly) whether the SNP is complete (converged or max_inner passes run).
// Fixed-order partial row sum: thread (r, j) adds rows r, r+R, ... (R = BLOCK / J), eight
// loads in flight at a time.  issue() only starts the first eight loads, so that the
// caller can queue other loads behind them (loads return in order: what is needed first
// must be issued first); finish() adds them up and walks the remaining rows.
template <int BLOCK>
struct RowSum {
  double t[8];
  const double *rows;
  uint32_t nrows, J, R, j, r;
  __device__ __forceinline__ void issue(const double *rows_, uint32_t nrows_, uint32_t J_) {
    rows = rows_;
    nrows = nrows_;
    J = J_;
    R = BLOCK / J;
    j = threadIdx.x % J;
    r = threadIdx.x / J;
    const uint32_t last = nrows > 0u ? nrows - 1u : 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = rows[(size_t)min(r + (uint32_t)u * R, last) * J + j];
  }
  __device__ __forceinline__ double finish() {
    double v = 0.0;
    if (r >= R) return v;
#pragma unroll
    for (int u = 0; u < 8; ++u) v += (r + (uint32_t)u * R < nrows) ? t[u] : 0.0;
    for (uint32_t g0 = r + 8u * R; g0 < nrows; g0 += 8u * R) {
      double s[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] = rows[(size_t)min(g0 + (uint32_t)u * R, nrows - 1u) * J + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += (g0 + (uint32_t)u * R < nrows) ? s[u] : 0.0;
    }
    return v;
  }
};

template <int BLOCK>
__device__ __forceinline__ double row_partial_sum(const double *rows, uint32_t nrows, uint32_t J) {
  RowSum<BLOCK> rs;
  rs.issue(rows, nrows, J);
  return rs.finish();
}

// Finish the pending pass described by `in` (load_pending: issued early, with the other state
// loads -- a load placed after the first barrier here would be a full memory latency on the
// critical path of every pass) from the per-thread partial row sums v (see
// RowSum; then r = 0..R-1): lambda_t[j] = eb_used[j] * sum_rows, then update_lambda +
// estimate_beta + convergence test (src/snpsamplinge.cc:356-364, :267-296; abs_mean
// src/matrix.hh:885-893).  Called by ALL threads of a workgroup.
// Outputs in LDS: s_lam = new lambda[loc], s_eb = new exp(Elogbeta[loc]).
// Returns (uniformly) whether the SNP is complete (converged or max_inner passes run).
struct PendingIn {  // what the epilogue needs from the previous launch's State, loaded by the
  double eb_used;   // caller together with the rest of the state (for j = tid < J): the
  double lam_old;   // exp(Elogbeta) the pass used, lambda before it, passes run so far
  uint32_t iters;
};
__device__ __forceinline__ PendingIn load_pending(const State *S, uint32_t J) {
  PendingIn in;
  const uint32_t tid = threadIdx.x;
  in.eb_used = S->eb[tid < J ? tid : 0u];
  in.lam_old = S->lam[tid < J ? tid : 0u];
  in.iters = S->iters;
  return in;
}
// The K x 2 epilogue proper, for thread tid < J holding lt = the row total of value tid:
// lambda_t = eb_used * lt, update_lambda, estimate_beta; outputs into LDS.  Called by threads < J.
__device__ __forceinline__ void epilogue_values(const DevParams &p, double lt, double eb_used, double lam_old,
                                                double *s_lam, double *s_eb, double *s_diff) {
  const uint32_t tid = threadIdx.x;
  // eta + b[k,t] * (row sum): the b factored out of the accumulation; an explicit fma so that
  // every kernel that inlines this rounds the same way whatever the compiler would contract
  const double nw = fma(lt, eb_used, (tid & 1u) ? p.eta1 : p.eta0);
  // exp(Elogbeta_kt) = exp(psi(lambda_kt) - psi(lambda_k0 + lambda_k1)) without a log: both
  // digammas in the split form z * exp(a) (tsamd_device.h), side by side in one instruction
  // stream; the pair sum comes from the neighbouring lane (t = 0/1 are adjacent threads)
  const double pair = nw + partner<1>(nw);
  double z1, a1, z2, a2;
  exp_digamma_split(nw, z1, a1);
  exp_digamma_split(pair, z2, a2);
  s_lam[tid] = nw;
  s_eb[tid] = (z1 * fast_rcp(z2)) * exp_nonpos(a1 - a2);
  s_diff[tid] = fabs(nw - lam_old);
}
// ... and the convergence decision every thread takes for itself after the barrier that follows:
// mean |dlambda| in the reference's order j = 0 .. J-1 (abs_mean, src/matrix.hh:885-893)
__device__ __forceinline__ bool epilogue_complete(const DevParams &p, uint32_t iters, uint32_t J, const double *s_diff) {
  double d = 0.0;
  for (uint32_t jj = 0; jj < J; ++jj) d += s_diff[jj];
  if ((J & (J - 1u)) == 0u)
    d *= 1.0 / (double)J;  // (a power of two: the same bits as the division, without its fifteen instructions)
  else
    d /= (double)J;
  return d < p.thresh || iters >= p.max_inner;
}

template <int BLOCK>
__device__ __forceinline__ bool finish_pending(const DevParams &p, const PendingIn &in, double v, uint32_t J,
                                               double *s_fin, double *s_lam, double *s_eb, double *s_diff) {
  const uint32_t tid = threadIdx.x;
  const uint32_t R = BLOCK / J;
  s_fin[tid] = v;
  __syncthreads();
  if (tid < J) {
    double lt = 0.0;
    for (uint32_t rr = 0; rr < R; ++rr) lt += s_fin[rr * J + tid];
    epilogue_values(p, lt, in.eb_used, in.lam_old, s_lam, s_eb, s_diff);
  }
  __syncthreads();
  return epilogue_complete(p, in.iters, J, s_diff);
}

// The SNP counters of Ctl, mirrored into the pinned host words (DevParams::host_error: [1] inner passes of the last
// completed SNP, [2] total passes, [3 + b] pass histogram bin b) by the one thread that publishes a SNP: the host reads
// them after a stream synchronise without a device-to-host copy (tsamd_snp_update, tsamd_total_passes, tsamd_pass_histogram).
__device__ __forceinline__ void count_snp(const DevParams &p, Ctl *ctl, uint32_t iters) {
  const uint32_t bin = min(iters, (uint32_t)TSAMD_PASS_HIST_BINS - 1u);
  ctl->last_iters = iters;
  ctl->total_passes += (unsigned long long)iters;
  ctl->pass_hist[bin] += 1ull;
  if (p.host_error) {
    __hip_atomic_store(p.host_error + 1, (unsigned long long)iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(p.host_error + 2, ctl->total_passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(p.host_error + 3 + bin, ctl->pass_hist[bin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// workgroup 0 publishes the completed SNP: final lambda / exp(Elogbeta) into the global
// arrays, counters, and the carried-forward state (eb stays the one the LAST executed
// pass used: the deferred gamma step needs it, src/snpsamplinge.cc:660-668).
__device__ __forceinline__ void publish_complete(const DevParams &p, Ctl *ctl, const State *S, State *W,
                                                 uint32_t J, const double *s_lam, const double *s_eb,
                                                 bool write_state) {
  const uint32_t tid = threadIdx.x;
  if (tid < J) {
    p.lam[(size_t)S->loc * J + tid] = s_lam[tid];
    p.eb[(size_t)S->loc * J + tid] = s_eb[tid];
    if (write_state) {
      W->lam[tid] = s_lam[tid];
      W->eb[tid] = S->eb[tid];
    }
  }
  if (tid == 0) {
    count_snp(p, ctl, S->iters);
    if (write_state) {
      W->idx = S->idx;
      W->valid = 1u;
      W->loc = S->loc;
      W->hol = S->hol;
      W->iters = S->iters;
      W->done = 1u;
      W->nrows = 0u;
      W->epoch = S->epoch + 1ull;
    }
  }
}

// the same for a kernel that ran several passes itself: everything explicit instead of read from S
__device__ __forceinline__ void publish_complete_v(const DevParams &p, Ctl *ctl, State *W, uint32_t J, uint32_t idx,
                                                   uint32_t loc, uint32_t hol, uint32_t iters, unsigned long long epoch_now,
                                                   double eb_last, const double *s_lam, const double *s_eb) {
  const uint32_t tid = threadIdx.x;
  if (tid < J) {
    p.lam[(size_t)loc * J + tid] = s_lam[tid];
    p.eb[(size_t)loc * J + tid] = s_eb[tid];
    W->lam[tid] = s_lam[tid];
    W->eb[tid] = eb_last;  // exp(Elogbeta) the LAST executed pass used (the deferred gamma step needs it)
  }
  if (tid == 0) {
    count_snp(p, ctl, iters);
    W->idx = idx;
    W->valid = 1u;
    W->loc = loc;
    W->hol = hol;
    W->iters = iters;
    W->done = 1u;
    W->nrows = 0u;
    W->epoch = epoch_now;
  }
}

__device__ __forceinline__ void carry_state(const State *S, State *W, uint32_t J) {
  const uint32_t tid = threadIdx.x;
  if (tid < J) {
    W->lam[tid] = S->lam[tid];
    W->eb[tid] = S->eb[tid];
  }
  if (tid == 0) {
    W->idx = S->idx;
    W->valid = S->valid;
    W->loc = S->loc;
    W->hol = S->hol;
    W->iters = S->iters;
    W->done = S->done;
    W->nrows = S->nrows;
    W->epoch = S->epoch + 1ull;
  }
}

// Sharded over the peer-to-peer exchange: wait (bounded) until all nflags partial rows of
// the previous launch (world ranks x its workgroups) have landed in this rank's buffer.
// Called by all threads of a workgroup.
__device__ __forceinline__ void wait_peer_rows(const DevParams &p, uint32_t slot, unsigned long long epoch,
                                               uint32_t nflags) {
  // a thread watches up to four flags at once (the flags are uncached: one poll is a trip to
  // memory, so the polls of one thread must not queue behind each other)
  const unsigned long long *seq = p.xchg->seq[slot];
  // a peer already timed out: do not spend another bounded wait in every later launch (the
  // results are void, tsamd_synchronize reports TSAMD_ECOMM)
  const bool dead = __hip_atomic_load(&p.xchg->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0ull;
  for (uint32_t t0 = threadIdx.x; t0 < nflags && !dead; t0 += 4u * blockDim.x) {
    const unsigned long long start = wall_clock64();  // 100 MHz
    while (true) {
      unsigned long long f[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t t = t0 + (uint32_t)u * blockDim.x;
        f[u] = __hip_atomic_load(&seq[t < nflags ? t : t0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (f[0] >= epoch && f[1] >= epoch && f[2] >= epoch && f[3] >= epoch) break;
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - start > 300000000ull) {  // 3 s: a peer died; report instead of hanging
        __hip_atomic_store(&p.xchg->error, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (p.host_error) __hip_atomic_store(p.host_error, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
}

// Sharded over the peer-to-peer exchange: workgroup 0 of EVERY launch of the sequence tells every
// rank that this rank has started launch `epoch_now` (tsamd_device.h, Xchg::prog).
__device__ __forceinline__ void publish_progress(const DevParams &p, unsigned long long epoch_now) {
  if (blockIdx.x == 0 && threadIdx.x < p.xchg_world)
    __hip_atomic_store(&p.peers[threadIdx.x]->prog[p.xchg_rank], epoch_now, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// ... and a launch about to store rows WITHOUT having waited for its peers' rows of the previous
// launch waits (bounded) until every peer has started the same launch: only then has every
// workgroup of every peer finished reading the slot that is about to be overwritten.  Threads
// 0 .. world-1 poll one flag each (local, uncached); the caller synchronises the workgroup.
__device__ __forceinline__ void wait_peer_progress(const DevParams &p, unsigned long long epoch_now) {
  if (threadIdx.x >= p.xchg_world || p.xchg_test_noguard != 0u) return;
  if (__hip_atomic_load(&p.xchg->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0ull) return;
  const unsigned long long start = wall_clock64();
  while (__hip_atomic_load(&p.xchg->prog[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < epoch_now) {
    __builtin_amdgcn_s_sleep(4);
    if (wall_clock64() - start > 300000000ull) {  // 3 s
      __hip_atomic_store(&p.xchg->error, epoch_now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (p.host_error) __hip_atomic_store(p.host_error, epoch_now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
  }
}

// test hook: stall a rank between its flag wait and its row reads (tests/test_gpu_multirank.py)
__device__ __forceinline__ void xchg_test_stall(const DevParams &p) {
  if (p.xchg_test_delay == 0u) return;
  const unsigned long long start = wall_clock64();
  while (wall_clock64() - start < (unsigned long long)p.xchg_test_delay) __builtin_amdgcn_s_sleep(8);
}

// End of a pass: the workgroup's partial row (value j = tid < J) leaves for the next launch --
// into this GPU's partial-row slot, or straight into every rank's exchange buffer followed by
// the epoch flag.  unguarded: the launch did not wait for its peers' previous rows (see above).
__device__ __forceinline__ void store_row(const DevParams &p, uint32_t par, unsigned long long epoch_now, double v,
                                          uint32_t J, double *rowsW, bool unguarded) {
  const uint32_t tid = threadIdx.x;
  if (p.xchg_world == 0u) {
    if (tid < J) rowsW[(size_t)blockIdx.x * J + tid] = v;  // read by the NEXT launch only
    return;
  }
  if (unguarded) {
    wait_peer_progress(p, epoch_now);
    __syncthreads();
  }
  if (tid < J) {
    // straight into every rank's exchange buffer (one 8-byte store per value and peer)
    const size_t at = ((size_t)p.xchg_rank * gridDim.x + blockIdx.x) * J + tid;
    for (uint32_t q = 0; q < p.xchg_world; ++q)
      __hip_atomic_store(&p.peers[q]->rows[par][at], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
  }
  __syncthreads();
  if (tid < p.xchg_world)
    __hip_atomic_store(&p.peers[tid]->seq[par][p.xchg_rank * gridDim.x + blockIdx.x], epoch_now, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// Partial row sum over the exchange buffer: the same fixed order as RowSum (thread (r, j) adds
// rows r, r+R, ...) with up to 32 loads in flight per thread -- the buffer is uncached, so
// every batch is a full trip to memory.
template <int BLOCK>
__device__ __forceinline__ double row_partial_sum_xchg(const double *rows, uint32_t nrows, uint32_t J) {
  const uint32_t R = BLOCK / J, j = threadIdx.x % J, r = threadIdx.x / J;
  double v = 0.0;
  if (r >= R || nrows == 0u) return v;
  for (uint32_t g0 = r; g0 < nrows; g0 += 32u * R) {
    double s[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) s[u] = rows[(size_t)min(g0 + (uint32_t)u * R, nrows - 1u) * J + j];
#pragma unroll
    for (int u = 0; u < 32; ++u) v += (g0 + (uint32_t)u * R < nrows) ? s[u] : 0.0;
  }
  return v;
}

// w[k] = exp(psi(g[k])) * exp(-a_max) = z_k * exp(a_k - a_max): Elogtheta exponentiated
// up to a per-individual constant, which cancels in phi (estimate_theta,
// src/snpsamplinge.cc:721-740).  The population with the largest a gets exp(0), so w never
// underflows for all k at once.
template <int KT>
__device__ __forceinline__ void gamma_to_w(const double (&g)[KT], double (&w)[KT]) {
  double z[KT], a[KT];
  double amax = -1.0e300;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    exp_digamma_split(g[k], z[k], a[k]);
    amax = fmax(amax, a[k]);
  }
#pragma unroll
  for (int k = 0; k < KT; ++k) w[k] = z[k] * exp_nonpos(a[k] - amax);
}

// SVI step for one individual (update_gamma + update_rho_indiv,
// src/snpsamplinge.cc:688-719) using phi recomputed from the weights w and the
// exp(Elogbeta) of the previous SNP's LAST pass (sb0/sb1).
template <int KT>
__device__ __forceinline__ void gamma_step_one(double (&g)[KT], const double (&w)[KT], const double (&sb0)[KT],
                                               const double (&sb1)[KT], double mom, double dad, uint32_t &c,
                                               const DevParams &p) {
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    s0 = fma(w[k], sb0[k], s0);
    s1 = fma(w[k], sb1[k], s1);
  }
  const double base = p.nodetau0 + (double)c;
  const double rho = (p.nodekappa == 0.5) ? 1.0 / sqrt(base) : pow(base, -p.nodekappa);
  c += 1u;
  const double c0 = mom * fast_rcp(s0), c1 = dad * fast_rcp(s1);
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const double e = c0 * (w[k] * sb0[k]) + c1 * (w[k] * sb1[k]);  // y*phi_mom + (2-y)*phi_dad
    g[k] += rho * (p.alpha + p.gamma_scale * e - g[k]);
  }
}

// VEC individuals per thread: 1 (8-byte row loads) or 2 (16-byte row loads).
template <int VEC>
struct Lanes;
template <>
struct Lanes<1> {
  using T = double;
  using C = uint32_t;
  static __device__ __forceinline__ void unpack(T v, double (&o)[1]) { o[0] = v; }
  static __device__ __forceinline__ T pack(const double (&o)[1]) { return o[0]; }
  static __device__ __forceinline__ void unpack_c(C v, uint32_t (&o)[1]) { o[0] = v; }
  static __device__ __forceinline__ C pack_c(const uint32_t (&o)[1]) { return o[0]; }
};
template <>
struct Lanes<2> {
  using T = double2;
  using C = uint2;
  static __device__ __forceinline__ void unpack(T v, double (&o)[2]) { o[0] = v.x; o[1] = v.y; }
  static __device__ __forceinline__ T pack(const double (&o)[2]) { return make_double2(o[0], o[1]); }
  static __device__ __forceinline__ void unpack_c(C v, uint32_t (&o)[2]) { o[0] = v.x; o[1] = v.y; }
  static __device__ __forceinline__ C pack_c(const uint32_t (&o)[2]) { return make_uint2(o[0], o[1]); }
};

// One inner pass.  KT == K exactly (one instantiation per K), so every k-loop is
// straight-line code and the K row loads of an iteration are issued back to back.
// Item i of a workgroup's chunk is VEC consecutive individuals: one 8*VEC-byte load per
// population row and 2*VEC bits of the 2-bit column.  The plain pass uses VEC = 2; the
// first pass, which also carries the gamma step, uses VEC = 1 to halve its registers.
#ifndef TSAMD_FIRST_WAVES
#define TSAMD_FIRST_WAVES 1
#endif
template <int KT, bool FIRST, int BLOCK, int VEC>
__global__ __launch_bounds__(BLOCK, (FIRST && VEC == 1) ? TSAMD_FIRST_WAVES : 1) void ts_pass(Ctl *ctl_a, double *partials_a, double *w_a, uint32_t npad_a, uint32_t chunk_a, uint32_t par_arg,
                                                                                           uint32_t nrows_hint, uint32_t local_rows_a, const DevParams p) {
  // The leading scalar arguments repeat what the kernel needs before anything else (control block,
  // partial rows, weight rows, geometry): built with -amdgpu-kernarg-preload-count they arrive in
  // SGPRs with the wave, so the first loads do not wait for a kernel-argument fetch; the rest of
  // the parameter block is fetched when it is first needed.
  // par_arg: bit 0 = launch parity (state / partial-row slot written), bit 1 = plain pass sweeps
  // its chunk backwards (set for the odd passes of a SNP, a property of the pass, not of the
  // launch parity: any cut of a schedule into calls or graphs gives the same summation order)
  const uint32_t par = par_arg & 1u;
  constexpr int kWaves = BLOCK / 64;
  using LN = Lanes<VEC>;
  using WT = typename LN::T;
  constexpr uint32_t kItemsPerWord = 16u / VEC;  // items per 32-bit word of the column
  constexpr uint32_t kCodeBits = 2u * VEC;
  constexpr uint32_t J = 2 * KT;
  __shared__ double s_eb[J];    // exp(Elogbeta) this pass uses
  __shared__ double s_lam[J];   // lambda[loc] before this pass' epilogue
  __shared__ double s_sb[J];    // FIRST: exp(Elogbeta) of the previous SNP's last pass
  __shared__ double s_plam[J];  // FIRST: final lambda / eb of the previous SNP when finished here
  __shared__ double s_peb[J];
  __shared__ double s_diff[J];
  __shared__ double s_red[kWaves][J];
  __shared__ double s_fin[BLOCK];

  Ctl *ctl = ctl_a;
  const State *S = &ctl->st[par ^ 1u];
  State *W = &ctl->st[par];
#ifdef TSAMD_WGTIME
  const unsigned long long wg_t0 = wall_clock64();
#endif
#ifdef TSAMD_TRACE
  unsigned long long tr[6];
  tr[0] = wall_clock64();
#define TSAMD_TR(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tr[k] = wall_clock64(); } while (0)
#else
#define TSAMD_TR(k) do { } while (0)
#endif
  // The state of the sequence first (scalar loads: everything below that matters waits for them).
  const uint32_t sidx = S->idx, svalid = S->valid, sdone = S->done, sloc = S->loc, shol = S->hol;
  const uint32_t siters = S->iters, snrows = S->nrows;
  const uint32_t sched_len = ctl->sched_len;
  const uint32_t *sched = ctl->sched;
  __builtin_amdgcn_sched_barrier(0);
  const uint32_t tid = threadIdx.x;
  const size_t np = npad_a;
  const uint32_t nitems = npad_a / VEC;
  const uint32_t chunk = chunk_a;
  const uint32_t begin = blockIdx.x * chunk;
  const uint32_t end = min(begin + chunk, nitems);

  auto load_rows = [&](uint32_t i, WT (&wv)[KT]) {
#pragma unroll
    for (int k = 0; k < KT; ++k) wv[k] = reinterpret_cast<const WT *>(w_a + (size_t)k * np)[i];
  };

  // Loads that do not depend on the state machine go first, so that they overlap its
  // dependent loads.  Order matters (loads return in order): the previous launch's partial
  // rows -- needed by the prologue; single GPU only; nrows_hint is that launch's grid size, a
  // launch-time constant -- then the first item's row data.
  const bool local_rows = local_rows_a != 0u;  // = p.xchg_world == 0 && p.rows_from_lt == 0
  RowSum<BLOCK> rowsum;
  // (first pass: only workgroup 0 needs the previous SNP's rows unless the slow path is taken)
  const bool rows_issued = local_rows && (!FIRST || blockIdx.x == 0);
  rowsum.issue(partials_a + (size_t)(par ^ 1u) * kMaxGrid * (2 * KT), rows_issued ? nrows_hint : 0u, 2 * KT);
  // Plain passes sweep their chunk forwards and backwards alternately (passes 2, 4, ... of a
  // SNP backwards): a pass starts on the addresses the previous one touched last (measured
  // 13.1 -> 11.1 us at N = 1M, K = 8; TSAMD_SWEEP=0 disables).
  WT bufA[KT];
  const uint32_t i0 = begin + tid;
  const uint32_t cnt = (i0 < end) ? (end - i0 + BLOCK - 1u) / BLOCK : 0u;  // this thread's items
  const bool rev = !FIRST && (par_arg & 2u) != 0u;  // (the host sets the bit only when alternating sweeps are on)
  auto item = [&](uint32_t t) { return rev ? i0 + (cnt - 1u - t) * BLOCK : i0 + t * BLOCK; };
  // What the epilogue needs (partial rows above, its per-thread inputs here) is requested before
  // the sweep's own data: loads return in order.  The sweep loads below are issued
  // unconditionally with clamped indices (a thread past the end of the last chunk re-reads the
  // array's last item): a load inside a branch would make every later wait conservative.
  const PendingIn pin = load_pending(S, J);
  auto item_or_last = [&](uint32_t t) { return cnt ? item(min(t, cnt - 1u)) : min(i0, nitems - 1u); };
  if constexpr (!FIRST) load_rows(item_or_last(0), bufA);
  __builtin_amdgcn_sched_barrier(0);
  // first pass: what it needs about the new SNP was captured one SNP ahead (NextSnp); both slots
  // are requested with the state, the one whose for_idx matches is used
  uint32_t nx_for[2] = {0xffffffffu, 0xffffffffu}, nx_ent[2] = {0u, 0u};
  double nx_lam[2] = {0.0, 0.0}, nx_eb[2] = {0.0, 0.0};
  if constexpr (FIRST) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      nx_for[q] = ctl->nxt[q].for_idx;
      nx_ent[q] = ctl->nxt[q].ent;
      nx_lam[q] = ctl->nxt[q].lam[tid < J ? tid : 0u];
      nx_eb[q] = ctl->nxt[q].eb[tid < J ? tid : 0u];
    }
  }
  // plain pass: the location is known as soon as the state is (it only changes in a first
  // pass), so the first column word is requested now and arrives during the epilogue
  uint32_t word_early = 0;
  if constexpr (!FIRST)
    word_early = reinterpret_cast<const uint32_t *>(p.bed + (size_t)sloc * p.colstride)[item_or_last(0) / kItemsPerWord];
  __builtin_amdgcn_sched_barrier(0);
  const double *rowsR = p.xchg_world  ? p.xchg->rows[par ^ 1u]
                        : p.rows_from_lt ? ctl->lt_sum[par ^ 1u]
                                         : partials_a + (size_t)(par ^ 1u) * kMaxGrid * J;
  const uint32_t nrowsR = p.xchg_world ? p.xchg_world * snrows : p.rows_from_lt ? 1u : snrows;
  double *rowsW = partials_a + (size_t)par * kMaxGrid * J;

#ifdef TSAMD_TRACE
  if (sidx == 12345678u) return;  // (forces the state load to complete before the stamp)
#endif
  TSAMD_TR(1);
  const bool pending = svalid != 0u && sdone == 0u;
  const unsigned long long epoch_now = S->epoch + 1ull;
  // First pass, fast path: the new SNP's entry and values come from NextSnp, and the previous
  // SNP's epilogue (row sum, update_lambda, estimate_beta) is only needed for publishing its final
  // values -- workgroup 0's job; every other workgroup goes straight to the sweep.  Slow path (the
  // first SNP of a schedule, the same location twice in a row, the end of the schedule): everybody
  // runs the epilogue and the values are read from the global arrays, as in a plain pass.
  int nx_sel = -1;
  if constexpr (FIRST) {
    const uint32_t nidx = sidx + 1u;
    nx_sel = nx_for[0] == nidx ? 0 : nx_for[1] == nidx ? 1 : -1;
    if (nx_sel >= 0 && (nidx >= sched_len || (nx_ent[nx_sel] & 0x7fffffffu) == sloc)) nx_sel = -1;
  }
  const bool first_fast = FIRST && nx_sel >= 0;
  const bool need_epilogue = pending && (!first_fast || blockIdx.x == 0);
  bool synced = false;  // sharded peer-to-peer: this workgroup has waited for its peers' previous rows
  if (p.xchg_world) {
    publish_progress(p, epoch_now);
    if (need_epilogue) {
      wait_peer_rows(p, par ^ 1u, S->epoch, nrowsR);
      xchg_test_stall(p);
      synced = true;
    }
  }
  double vrow = 0.0;
  if (need_epilogue)
    vrow = (rows_issued && nrowsR == nrows_hint) ? rowsum.finish()
           : p.xchg_world                        ? row_partial_sum_xchg<BLOCK>(rowsR, nrowsR, J)
                                                 : row_partial_sum<BLOCK>(rowsR, nrowsR, J);
  TSAMD_TR(2);
  uint32_t loc, hol, idx, iters;
  bool do_gamma = false;
  uint32_t prev_loc = 0;

  if constexpr (!FIRST) {
    if (!pending) {  // nothing in flight (converged earlier, or schedule exhausted): carry state
      if (blockIdx.x == 0) carry_state(S, W, J);
      return;
    }
    const bool complete = finish_pending<BLOCK>(p, pin, vrow, J, s_fin, s_lam, s_eb, s_diff);
    if (complete) {
      if (blockIdx.x == 0) publish_complete(p, ctl, S, W, J, s_lam, s_eb, true);
      return;
    }
    loc = sloc;
    hol = shol;
    idx = sidx;
    iters = siters + 1u;
  } else {
    idx = sidx + 1u;  // 0xffffffff + 1 = 0: first SNP of the schedule
    if (need_epilogue) finish_pending<BLOCK>(p, pin, vrow, J, s_fin, s_plam, s_peb, s_diff);
    if (idx >= sched_len) {  // schedule exhausted: complete what is pending, carry state
      if (blockIdx.x == 0) {
        if (pending)
          publish_complete(p, ctl, S, W, J, s_plam, s_peb, true);
        else
          carry_state(S, W, J);
      }
      return;
    }
    iters = 1u;
    do_gamma = svalid != 0u && shol == 0u;
    prev_loc = sloc;
    if (first_fast) {
      const uint32_t ent = nx_sel ? nx_ent[1] : nx_ent[0];
      loc = ent & 0x7fffffffu;
      hol = ent >> 31;
      if (tid < J) {
        s_sb[tid] = pin.eb_used;  // = S->eb[tid]
        s_lam[tid] = nx_sel ? nx_lam[1] : nx_lam[0];
        s_eb[tid] = nx_sel ? nx_eb[1] : nx_eb[0];
      }
    } else
    {
      const uint32_t ent = sched[idx];
      loc = ent & 0x7fffffffu;
      hol = ent >> 31;
      if (tid < J) {
        s_sb[tid] = S->eb[tid];
        if (pending && sloc == loc) {  // same location twice in a row: its final values are still local
          s_lam[tid] = s_plam[tid];
          s_eb[tid] = s_peb[tid];
        } else {
          s_lam[tid] = p.lam[(size_t)loc * J + tid];
          s_eb[tid] = p.eb[(size_t)loc * J + tid];
        }
      }
    }
    __syncthreads();
  }

  double b0[KT], b1[KT];  // wave-uniform: kept in SGPRs
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    b0[k] = uniform_f64(s_eb[2 * k]);
    b1[k] = uniform_f64(s_eb[2 * k + 1]);
  }
  double acc0[KT], acc1[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) acc0[k] = acc1[k] = 0.0;

  const uint32_t *col = reinterpret_cast<const uint32_t *>(p.bed + (size_t)loc * p.colstride);
  const uint32_t *pcol = reinterpret_cast<const uint32_t *>(p.bed + (size_t)prev_loc * p.colstride);

  TSAMD_TR(3);
  // ---- main sweep -------------------------------------------------------------------
  auto accumulate = [&](uint32_t i, const double (&w)[VEC][KT], uint32_t word) {
    const uint32_t code = word >> (kCodeBits * (i % kItemsPerWord));
    double c0[VEC], c1[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      double mom, dad;
      bool ok;
      code_weights((code >> (2 * v)) & 3u, mom, dad, ok);
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        s0 = fma(w[v][k], b0[k], s0);
        s1 = fma(w[v][k], b1[k], s1);
      }
      // (the first pass is arithmetic-bound next to its gamma step: reciprocal + two Newton steps
      // there, IEEE division in the bandwidth-bound plain pass)
      c0[v] = FIRST ? mom * fast_rcp(s0) : mom / s0;
      c1[v] = FIRST ? dad * fast_rcp(s1) : dad / s1;
    }
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
      for (int v = VEC - 1; v >= 0; --v) {
        acc0[k] = fma(c0[v], w[v][k], acc0[k]);
        acc1[k] = fma(c1[v], w[v][k], acc1[k]);
      }
  };
  auto unpack_rows = [&](const WT (&wv)[KT], double (&w)[VEC][KT]) {
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      double t[VEC];
      LN::unpack(wv[k], t);
#pragma unroll
      for (int v = 0; v < VEC; ++v) w[v][k] = t[v];
    }
  };

  if constexpr (!FIRST) {
    // two-stage software pipeline: the next item's K row loads are in flight while the
    // current one is reduced (prefetch addresses are clamped, not predicated, so the
    // load/wait counts are static)
    auto consume = [&](uint32_t i, const WT (&wv)[KT], uint32_t word) {
      double w[VEC][KT];
      unpack_rows(wv, w);
      accumulate(i, w, word);
    };
    WT bufB[KT];
    uint32_t wordA = 0, wordB = 0;
    if (cnt) {
      wordA = word_early;
      uint32_t t = 0;
      while (true) {
        const uint32_t t1 = t + 1u;
        const uint32_t j1 = item(t1 < cnt ? t1 : t);
        load_rows(j1, bufB);
        wordB = col[j1 / kItemsPerWord];
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the arithmetic
        consume(item(t), bufA, wordA);
        __builtin_amdgcn_sched_barrier(0);
        if (t1 >= cnt) break;
        const uint32_t t2 = t + 2u;
        const uint32_t j2 = item(t2 < cnt ? t2 : t1);
        load_rows(j2, bufA);
        wordA = col[j2 / kItemsPerWord];
        __builtin_amdgcn_sched_barrier(0);
        consume(item(t1), bufB, wordB);
        __builtin_amdgcn_sched_barrier(0);
        if (t2 >= cnt) break;
        t = t2;
      }
    }
  } else {
    double sb0[KT], sb1[KT];  // exp(Elogbeta) of the previous SNP's last pass (wave-uniform)
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      // (kept in vector registers: with b0/b1 and the polynomial constants they do not fit the
      // scalar file, and a spilled scalar costs a v_readlane per use)
      sb0[k] = do_gamma ? s_sb[2 * k] : 0.0;
      sb1[k] = do_gamma ? s_sb[2 * k + 1] : 0.0;
    }
    // software pipeline: the next item's rows (weights, gamma, counters, column words) are
    // requested before the current item's transcendental-heavy update starts
    using CT = typename LN::C;
    auto load_item = [&](uint32_t i, WT (&wv)[KT], WT (&gv)[KT], CT &cv, uint32_t &word, uint32_t &pword) {
      load_rows(i, wv);
      word = col[i / kItemsPerWord];
      if (do_gamma) {
#pragma unroll
        for (int k = 0; k < KT; ++k) gv[k] = reinterpret_cast<const WT *>(p.gam + (size_t)k * np)[i];
        cv = reinterpret_cast<const CT *>(p.cnt)[i];
        pword = pcol[i / kItemsPerWord];
      }
    };
    WT wv[KT], gv[KT], wv_n[KT], gv_n[KT];
    CT cv{}, cv_n{};
    uint32_t word = 0, pword = 0, word_n = 0, pword_n = 0;
    if (i0 < end) load_item(i0, wv, gv, cv, word, pword);
    for (uint32_t i = i0; i < end; i += BLOCK) {
      const uint32_t inext = (i + BLOCK < end) ? i + BLOCK : i;  // clamped: static load counts
      load_item(inext, wv_n, gv_n, cv_n, word_n, pword_n);
      __builtin_amdgcn_sched_barrier(0);
      double w[VEC][KT];
      if (do_gamma) {
        const uint32_t pcode = pword >> (kCodeBits * (i % kItemsPerWord));
        unpack_rows(wv, w);
        double g[VEC][KT];
        unpack_rows(gv, g);
        uint32_t cn[VEC];
        LN::unpack_c(cv, cn);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          double mom, dad;
          bool ok;
          code_weights((pcode >> (2 * v)) & 3u, mom, dad, ok);
          if (ok) {
#if !defined(TSAMD_ABL) || TSAMD_ABL == 1 || TSAMD_ABL == 3
            gamma_step_one<KT>(g[v], w[v], sb0, sb1, mom, dad, cn[v], p);
#endif
#if !defined(TSAMD_ABL) || TSAMD_ABL == 3
            gamma_to_w<KT>(g[v], w[v]);
#endif
          }
        }
#if defined(TSAMD_ABL) && TSAMD_ABL == 3
        if (w[0][0] == 123.456)
#endif
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          double tg[VEC], tw[VEC];
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            tg[v] = g[v][k];
            tw[v] = w[v][k];
          }
          reinterpret_cast<WT *>(p.gam + (size_t)k * np)[i] = LN::pack(tg);
          reinterpret_cast<WT *>(w_a + (size_t)k * np)[i] = LN::pack(tw);
        }
        reinterpret_cast<CT *>(p.cnt)[i] = LN::pack_c(cn);
      } else {
        unpack_rows(wv, w);
      }
      accumulate(i, w, word);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        wv[k] = wv_n[k];
        gv[k] = gv_n[k];
      }
      cv = cv_n;
      word = word_n;
      pword = pword_n;
    }
  }

  TSAMD_TR(4);
  // workgroup reduction, fixed order: lanes (halving butterfly) -> waves (0..kWaves-1)
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  {
    using Fold = WaveFold<2 * KT>;
    double v[Fold::P];
#pragma unroll
    for (int q = 0; q < Fold::P; ++q) v[q] = 0.0;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      v[2 * k] = acc0[k];
      v[2 * k + 1] = acc1[k];
    }
    const double tot = Fold::fold(v, lane);
    const int slot = Fold::slot(lane);
    constexpr uint32_t kRep = 64 / Fold::P;  // lanes sharing one slot
    if ((lane & (kRep - 1u)) == 0u && slot < (int)J) s_red[wave][slot] = tot;
  }
  __syncthreads();
  {
    double v = 0.0;
    if (tid < J) {
      v = s_red[0][tid];
#pragma unroll
      for (int wv = 1; wv < kWaves; ++wv) v += s_red[wv][tid];
    }
    store_row(p, par, epoch_now, v, J, rowsW, !synced);
  }

  // workgroup 0 publishes the state the next launch starts from
  if (blockIdx.x == 0) {
    if (FIRST && pending) publish_complete(p, ctl, S, W, J, s_plam, s_peb, false);
    if constexpr (FIRST) {  // ... and captures what the first pass of the next SNP will need
      NextSnp *NW = &ctl->nxt[idx & 1u];
      const uint32_t nidx = idx + 1u;
      if (nidx < sched_len) {
        const uint32_t ent2 = sched[nidx];
        const uint32_t loc2 = ent2 & 0x7fffffffu;
        if (tid < J) {
          const bool local = pending && loc2 == sloc;  // just finished here: not yet readable from the arrays
          NW->lam[tid] = local ? s_plam[tid] : p.lam[(size_t)loc2 * J + tid];
          NW->eb[tid] = local ? s_peb[tid] : p.eb[(size_t)loc2 * J + tid];
        }
        if (tid == 0) {
          NW->ent = ent2;
          NW->for_idx = nidx;
        }
      } else if (tid == 0) {
        NW->for_idx = 0xffffffffu;
      }
    }
    if (tid < J) {
      W->lam[tid] = s_lam[tid];
      W->eb[tid] = s_eb[tid];
    }
    if (tid == 0) {
      W->idx = idx;
      W->valid = 1u;
      W->loc = loc;
      W->hol = hol;
      W->iters = iters;
      W->done = 0u;
      W->nrows = gridDim.x;
      W->epoch = epoch_now;
    }
  }
#ifdef TSAMD_WGTIME
  {  // start / finish stamps of the workgroups of one first pass and one plain pass, kept in the unused tail of
     // the partial-row buffer and printed by workgroup 0 of the next first pass
    unsigned long long *stamps = reinterpret_cast<unsigned long long *>(partials_a + (size_t)2 * kMaxGrid * 2 * TSAMD_MAX_K) - 4 * kMaxGrid;
    if (tid == 0 && idx == 41u && (FIRST || iters == 5u)) {
      stamps[(FIRST ? 0 : 2) * kMaxGrid + blockIdx.x] = wall_clock64();
      stamps[(FIRST ? 1 : 3) * kMaxGrid + blockIdx.x] = wg_t0;
    }
    if (FIRST && tid == 0 && idx == 42u && blockIdx.x == 0) {
      for (uint32_t b = 0; b < gridDim.x; ++b) printf("wgtime %u %llu %llu\n", b, stamps[kMaxGrid + b], stamps[b]);
      for (uint32_t b = 0; b < 256u; ++b) printf("wgplain %u %llu %llu\n", b, stamps[3 * kMaxGrid + b], stamps[2 * kMaxGrid + b]);
    }
  }
#endif
#ifdef TSAMD_TRACE
  TSAMD_TR(5);
  if (blockIdx.x == 0 && tid == 0 && idx >= 40u && idx < 44u)
    printf("trace %s idx %u it %u: state %llu rows %llu epilogue %llu sweep %llu tail %llu (x10 ns)\n", FIRST ? "first" : "plain",
           idx, iters, tr[1] - tr[0], tr[2] - tr[1], tr[3] - tr[2], tr[4] - tr[3], tr[5] - tr[4]);
#endif
}


// ---------------------------------------------------------------------------
// ts_resident<K>: ALL plain passes of a SNP in one launch (single GPU, K <= 8, at most eight items
// per thread, i.e. N <= ~1M: the shard's weights fit the register file -- 256 CUs x 256 threads x
// 8 items x K rows x 16 bytes = 64 MB at K = 8).  The first of its passes streams the weights
// exactly like ts_pass<K,false,512,2> and keeps them; every later pass runs from registers.
// Between passes the workgroups exchange their partial rows inside the launch (ResXchg,
// tsamd_device.h): measured 3.1 us per exchange (tools/xchg_probe.hip), about what a kernel
// boundary + state reload + row reads cost -- so the gain is the 5 us weight sweep of every pass
// after the first.  Same state machine as the launch-per-pass sequence: it starts from the
// first pass' State and partial rows and leaves State / partial rows for the next first pass (or
// ts_flush); the workgroups reach the complete / continue decision identically from the same
// totals.  Needs every workgroup resident at once (grid <= CUs, checked by the host); every wait
// is bounded and a failure ends all later waits (reported by tsamd_synchronize).
constexpr int kResidentMaxK = 8;    // ts_resident holds kResidentItems items x K rows x 16 bytes per thread in registers
constexpr int kResidentItems = 8;   // (256-thread workgroups, one per compute unit)
constexpr int kResidentBlock = 256;

template <int N>
__device__ __forceinline__ bool res_sweep(const unsigned long long *base, uint32_t tag, uint32_t nvalid_gran, uint32_t group,
                                          uint32_t grid, bool by_member, unsigned (&v)[N], unsigned long long *abort_word,
                                          unsigned long long *host_flag, uint32_t lane) {
  // lane l, load i: granule c = l % 32 of row 2 i + (l >= 32); a row is a member's (level 1: it
  // exists if member * 8 + group < grid) or a group's (level 2: group index < min(grid, 8))
  const uint32_t c = lane & 31u;
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const uint32_t row = 2u * (uint32_t)i + (lane >> 5);
      const bool exists = c < nvalid_gran && (by_member ? row * (uint32_t)kResGroups + group < grid : row < min(grid, (uint32_t)kResGroups));
      const unsigned long long x = __hip_atomic_load(base + lane + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v[i] = exists ? (unsigned)x : 0u;
      ok &= !exists || (unsigned)(x >> 32) == tag;
    }
    if (__all(ok)) return true;
    if (wall_clock64() - t0 > 300000000ull ||  // 3 s
        __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) {
      if (lane == 0) {
        __hip_atomic_store(abort_word, (unsigned long long)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (host_flag) __hip_atomic_store(host_flag, (unsigned long long)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// sum of the N rows a wave has swept (lo / hi halves of a double sit in neighbouring lanes): on
// return lane 2 j (j < J) holds the total of value j over rows 0, 2, 4, ... plus rows 1, 3, 5, ...
template <int N>
__device__ __forceinline__ double res_sum(const unsigned (&v)[N], uint32_t lane) {
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const unsigned other = (unsigned)__shfl_xor((int)v[i], 1);
    const unsigned lo = (lane & 1u) ? other : v[i], hi = (lane & 1u) ? v[i] : other;
    s += __longlong_as_double(((unsigned long long)hi << 32) | lo);
  }
  return s + __shfl_xor(s, 32);
}

// Level 2 across ranks (ts_schedule, sharded): N row pairs of this rank's Xchg::res_sums slot, written by the group
// leaders of all ranks with system-scope stores; rows >= nrows (world * 8) do not exist.  Returns the fixed-order
// total in lanes 2 j like res_sum; false when the bounded wait gave up.
template <int N>
__device__ __forceinline__ bool res_sweep_ranks(const unsigned long long *base, uint32_t tag, uint32_t nvalid_gran, uint32_t nrows,
                                                double &total, unsigned long long *abort_word, unsigned long long *host_flag,
                                                uint32_t lane) {
  const uint32_t c = lane & 31u;
  const unsigned long long t0 = wall_clock64();
  unsigned v[N];
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const uint32_t row = 2u * (uint32_t)i + (lane >> 5);
      const bool exists = c < nvalid_gran && row < nrows;
      const unsigned long long x = __hip_atomic_load(base + lane + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      v[i] = exists ? (unsigned)x : 0u;
      ok &= !exists || (unsigned)(x >> 32) == tag;
    }
    if (__all(ok)) break;
    if (wall_clock64() - t0 > 300000000ull ||  // 3 s
        __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) {
      if (lane == 0) {
        __hip_atomic_store(abort_word, (unsigned long long)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (host_flag) __hip_atomic_store(host_flag, (unsigned long long)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      total = 0.0;
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  total = res_sum<N>(v, lane);
  return true;
}

template <int KT>
__global__ __launch_bounds__(256, 1) void ts_resident(Ctl *ctl_a, double *partials_a, double *w_a, uint32_t npad_a, uint32_t chunk_a,
                                                      uint32_t par_arg, uint32_t nrows_hint, ResXchg *xb, const DevParams p) {
  // 256 threads, one wave per SIMD: the whole 512-register file per lane is this wave's (8 items x K
  // rows x 4 registers = 256 at K = 8, plus accumulators and temporaries)
  constexpr int BLOCK = 256, kWaves = BLOCK / 64, kItems = kResidentItems;
  using WT = double2;
  constexpr uint32_t kItemsPerWord = 8u, kCodeBits = 4u;
  constexpr uint32_t J = 2 * KT;
  static_assert(2 * J <= (uint32_t)kResGran, "row does not fit the exchange granules");
  __shared__ double s_eb[J], s_lam[J], s_diff[J], s_tot[J];
  __shared__ double s_red[kWaves][J];
  __shared__ double s_fin[BLOCK];
  __shared__ int s_alive;

  const uint32_t par = par_arg & 1u;
  Ctl *ctl = ctl_a;
  const State *S = &ctl->st[par ^ 1u];
  State *W = &ctl->st[par];
  const uint32_t sidx = S->idx, svalid = S->valid, sdone = S->done, sloc = S->loc, shol = S->hol;
  const uint32_t siters = S->iters, snrows = S->nrows;
  const uint32_t xseq0 = ctl->xseq;  // (workgroup 0 advances it when it leaves, after everybody's first exchange)
  __builtin_amdgcn_sched_barrier(0);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const size_t np = npad_a;
  const uint32_t nitems = npad_a / 2u;
  const uint32_t begin = blockIdx.x * chunk_a, end = min(begin + chunk_a, nitems);
  const uint32_t i0 = begin + tid;
  const uint32_t cnt = (i0 < end) ? min((end - i0 + BLOCK - 1u) / BLOCK, (uint32_t)kItems) : 0u;
  auto item_or_last = [&](uint32_t t) { return cnt ? i0 + min(t, cnt - 1u) * BLOCK : min(i0, nitems - 1u); };
  auto load_rows = [&](uint32_t i, WT (&wv)[KT]) {
#pragma unroll
    for (int k = 0; k < KT; ++k) wv[k] = reinterpret_cast<const WT *>(w_a + (size_t)k * np)[i];
  };

  // the first pass' partial rows and the epilogue's inputs first, then the first item's rows
  RowSum<BLOCK> rowsum;
  rowsum.issue(partials_a + (size_t)(par ^ 1u) * kMaxGrid * J, nrows_hint, J);
  const PendingIn pin = load_pending(S, J);
  WT buf[kItems][KT];
  load_rows(item_or_last(0), buf[0]);
  __builtin_amdgcn_sched_barrier(0);
  uint32_t word[kItems];
  {
    const uint32_t *col = reinterpret_cast<const uint32_t *>(p.bed + (size_t)sloc * p.colstride);
#pragma unroll
    for (int t = 0; t < kItems; ++t) word[t] = col[item_or_last((uint32_t)t) / kItemsPerWord];
  }
  __builtin_amdgcn_sched_barrier(0);

  const bool pending = svalid != 0u && sdone == 0u;
  const unsigned long long epoch_now = S->epoch + 1ull;
  if (!pending) {  // nothing in flight (schedule exhausted, dry replay): carry the state forward
    if (blockIdx.x == 0) carry_state(S, W, J);
    return;
  }
  const double vrow = (snrows == nrows_hint) ? rowsum.finish()
                                             : row_partial_sum<BLOCK>(partials_a + (size_t)(par ^ 1u) * kMaxGrid * J, snrows, J);
  bool complete = finish_pending<BLOCK>(p, pin, vrow, J, s_fin, s_lam, s_eb, s_diff);
  uint32_t iters = siters;
  double eb_used = pin.eb_used;  // (threads < J) exp(Elogbeta) the last executed pass used
  const uint32_t g = blockIdx.x % (uint32_t)kResGroups, m = blockIdx.x / (uint32_t)kResGroups;
  double lam_old = 0.0;
  double b0[KT], b1[KT], acc0[KT], acc1[KT];
  // start of a pass: the values the previous epilogue left in LDS become this pass' inputs
  auto begin_pass = [&]() {
    iters += 1u;
    lam_old = s_lam[tid < J ? tid : 0u];
    eb_used = s_eb[tid < J ? tid : 0u];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      b0[k] = uniform_f64(s_eb[2 * k]);
      b1[k] = uniform_f64(s_eb[2 * k + 1]);
      acc0[k] = acc1[k] = 0.0;
    }
  };
  auto consume = [&](uint32_t i, const WT (&wv)[KT], uint32_t wd) {
    const uint32_t code = wd >> (kCodeBits * (i % kItemsPerWord));
    double c0[2], c1[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      double mom, dad;
      bool ok;
      code_weights((code >> (2 * v)) & 3u, mom, dad, ok);
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const double wk = v ? wv[k].y : wv[k].x;
        s0 = fma(wk, b0[k], s0);
        s1 = fma(wk, b1[k], s1);
      }
      c0[v] = mom * fast_rcp(s0);  // (arithmetic-bound from the second sweep on)
      c1[v] = dad * fast_rcp(s1);
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      acc0[k] = fma(c0[1], wv[k].y, acc0[k]);
      acc1[k] = fma(c1[1], wv[k].y, acc1[k]);
      acc0[k] = fma(c0[0], wv[k].x, acc0[k]);
      acc1[k] = fma(c1[0], wv[k].x, acc1[k]);
    }
  };
  // end of a pass: workgroup reduction; then either (the cap) hand rows and state to the next launch,
  // or exchange the rows inside the launch and run the epilogue.  Returns true when the kernel is over.
  uint32_t xcount = 0u;  // exchanges of this launch
  auto finish_pass = [&]() -> bool {
    {
      using Fold = WaveFold<2 * KT>;  // fixed order: lanes (halving butterfly) -> waves
      double v[Fold::P];
#pragma unroll
      for (int q = 0; q < Fold::P; ++q) v[q] = 0.0;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        v[2 * k] = acc0[k];
        v[2 * k + 1] = acc1[k];
      }
      const double tot = Fold::fold(v, lane);
      const int slot = Fold::slot(lane);
      constexpr uint32_t kRep = 64 / Fold::P;
      if ((lane & (kRep - 1u)) == 0u && slot < (int)J) s_red[wave][slot] = tot;
    }
    __syncthreads();
    double row = 0.0;
    if (tid < J) {
      row = s_red[0][tid];
#pragma unroll
      for (int wv = 1; wv < kWaves; ++wv) row += s_red[wv][tid];
    }
    if (iters >= p.max_inner) {
      // the cap: this was the SNP's last pass; its rows and the state go to the next launch (a first
      // pass or ts_flush), exactly as the last plain pass of the launch-per-pass sequence leaves them
      if (tid < J) (partials_a + (size_t)par * kMaxGrid * J)[(size_t)blockIdx.x * J + tid] = row;
      if (blockIdx.x == 0) {
        if (tid < J) {
          W->lam[tid] = lam_old;
          W->eb[tid] = eb_used;
        }
        if (tid == 0) {
          W->idx = sidx;
          W->valid = 1u;
          W->loc = sloc;
          W->hol = shol;
          W->iters = iters;
          W->done = 0u;
          W->nrows = gridDim.x;
          W->epoch = epoch_now;
          ctl->xseq = xseq0 + xcount;
        }
      }
      return true;
    }
    // ---- in-launch exchange of the partial rows ---------------------------------------------
    xcount += 1u;
    const uint32_t tag = xseq0 + xcount;
    if (tid < J) {
      const unsigned long long bits = __double_as_longlong(row);
      __hip_atomic_store(&xb->rows[g][m][2 * tid], ((unsigned long long)tag << 32) | (uint32_t)bits, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&xb->rows[g][m][2 * tid + 1], ((unsigned long long)tag << 32) | (uint32_t)(bits >> 32), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave == 0) {
      bool alive = true;
      if (m == 0) {  // leader of group g
        unsigned v[kResMembers / 2];
        alive = res_sweep<kResMembers / 2>(&xb->rows[g][0][0], tag, 2 * J, g, gridDim.x, true, v, &xb->abort_word, p.host_error, lane);
        const double s = res_sum<kResMembers / 2>(v, lane);
        if (lane < 2 * J && !(lane & 1u)) {
          const unsigned long long bits = __double_as_longlong(s);
          __hip_atomic_store(&xb->sums[tag & 1u][g][lane], ((unsigned long long)tag << 32) | (uint32_t)bits, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&xb->sums[tag & 1u][g][lane + 1], ((unsigned long long)tag << 32) | (uint32_t)(bits >> 32),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      unsigned v2[kResGroups / 2];
      alive = res_sweep<kResGroups / 2>(&xb->sums[tag & 1u][0][0], tag, 2 * J, 0u, gridDim.x, false, v2, &xb->abort_word, p.host_error, lane) && alive;
      const double s = res_sum<kResGroups / 2>(v2, lane);
      if (lane < 2 * J && !(lane & 1u)) s_tot[lane >> 1] = s;
      if (lane == 0) s_alive = alive ? 1 : 0;
    }
    __syncthreads();
    if (!s_alive) return true;  // (the abort word is set: tsamd_synchronize reports it; the state is void)
    if (tid < J) epilogue_values(p, s_tot[tid], eb_used, lam_old, s_lam, s_eb, s_diff);
    __syncthreads();
    complete = epilogue_complete(p, iters, J, s_diff);
    return false;
  };
  auto publish = [&]() {
    if (blockIdx.x == 0) {
      publish_complete_v(p, ctl, W, J, sidx, sloc, shol, iters, epoch_now, eb_used, s_lam, s_eb);
      if (tid == 0) ctl->xseq = xseq0 + xcount;
    }
  };

  if (complete) {  // (the first pass was the SNP's last)
    publish();
    return;
  }
  // first sweep: stream the weights like the plain pass (the next two items' rows in flight while the
  // current one is reduced; clamped, unconditional loads) -- and keep them
  begin_pass();
  load_rows(item_or_last(1u), buf[1]);
#pragma unroll
  for (int t = 0; t < kItems; ++t) {
    if (t + 2 < kItems) load_rows(item_or_last((uint32_t)t + 2u), buf[t + 2]);
    __builtin_amdgcn_sched_barrier(0);
    if ((uint32_t)t < cnt) consume(i0 + (uint32_t)t * BLOCK, buf[t], word[t]);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (finish_pass()) return;
  // every later sweep runs from registers (one item at a time: interleaving them would only
  // multiply the temporaries)
  for (;;) {
    if (complete) {
      publish();
      return;
    }
    begin_pass();
#pragma unroll
    for (int t = 0; t < kItems; ++t) {
      if ((uint32_t)t < cnt) consume(i0 + (uint32_t)t * BLOCK, buf[t], word[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (finish_pass()) return;
  }
}


// ---------------------------------------------------------------------------
// ts_schedule<K, PARTIAL, WR>: a WHOLE schedule in one launch (K <= 8, shards up to ~1M individuals per GPU: the
// residency conditions of ts_resident; one GPU, or -- WR > 0 -- one launch per rank of a sharded run).  The weights w
// stay in registers from the first SNP to the last: the gamma step of a SNP reads and writes gamma (and c_n) only --
// half of it from LDS at K = 8 -- and overwrites the registers with the new weights, every pass runs from registers,
// and every pass ends with the in-launch exchange of the partial rows (ResXchg; level 2 across the ranks through
// Xchg::res_sums when sharded).  Per SNP the memory traffic drops from (I + 3) 8NK to 8NK .. 16NK (gamma read +
// write); w and the LDS-resident gamma are written back once, at the end of the launch.
// Same semantics as the launch-per-pass state machine: starts from the State the previous call left (its last SNP
// complete, its gamma step possibly pending) and leaves such a State.  PARTIAL: skip the item bodies no thread of
// the workgroup owns (small shards).  WR: row pairs per lane of the cross-rank level 2 (0: single GPU).
// ts_schedule: how many of a thread's items keep their gamma in LDS (per item: K rows x 16 bytes + c_n for 256 threads;
// 4 KB of the 160 KB stay free for the small arrays), and which streamed item follows item t (items = none)
constexpr int sched_lds_items(int k, int items) {
#ifdef TSAMD_SCHED_LDS_ITEMS  // (experiments, tools/variant.sh)
  return TSAMD_SCHED_LDS_ITEMS < items ? TSAMD_SCHED_LDS_ITEMS : items;
#else
  const int per_item = (k * 16 + 8) * 256, n = (160 * 1024 - 4096) / per_item;
  return n < items ? n : items;
#endif
}
constexpr int sched_next_streamed(int t, int k, int items) {
  const int lds = sched_lds_items(k, items);
  for (int u = t + 1; u < items; ++u)
    if (((u + 1) * lds) / items == (u * lds) / items) return u;
  return items;
}

template <int KT, bool PARTIAL, int WR>
__global__ __launch_bounds__(256, 1) void ts_schedule(Ctl *ctl_a, double *w_a, uint32_t npad_a, uint32_t chunk_a, uint32_t par_arg,
                                                      const uint32_t *sched, uint32_t n_sched, ResXchg *xb, const DevParams p) {
  constexpr int BLOCK = 256, kWaves = BLOCK / 64, kItems = kResidentItems;
  using WT = double2;
  constexpr uint32_t kItemsPerWord = 8u, kCodeBits = 4u;
  constexpr uint32_t J = 2 * KT;
  __shared__ double s_eb[J], s_lam[J], s_diff[J], s_tot[J], s_sb[J], s_plam[J], s_peb[J];
  __shared__ double s_red[kWaves][J];
  __shared__ int s_alive;
  // gamma (and c_n) of kLds of a thread's eight items stay in LDS for the whole launch -- at K = 8 that is 4 x 32 KB of the
  // 160 KB, every second item; at K <= 4 all of them -- so the gamma step streams only the others from memory, one item ahead
  // and spread evenly over the step.  Memory sees them again when the launch ends.
  constexpr int kLds = sched_lds_items(KT, kItems);
  __shared__ WT s_gam[kLds > 0 ? kLds : 1][KT][BLOCK];
  __shared__ uint2 s_cn[kLds > 0 ? kLds : 1][BLOCK];
  auto is_lds = [](int t) { return ((t + 1) * kLds) / kItems != (t * kLds) / kItems; };
  auto lds_slot = [](int t) { return (t * kLds) / kItems; };
  const uint32_t par = par_arg & 1u;
  Ctl *ctl = ctl_a;
  const State *S = &ctl->st[par ^ 1u];
  State *W = &ctl->st[par];
  const uint32_t svalid = S->valid, sloc = S->loc, shol = S->hol, siters = S->iters, sidx = S->idx;
  const unsigned long long epoch_now = S->epoch + 1ull;
  const uint32_t xseq0 = ctl->xseq;  // (workgroup 0 advances it when it leaves, after everybody's first exchange)
  uint32_t tid = threadIdx.x;
  const size_t np = npad_a;
  const uint32_t nitems = npad_a / 2u;
  const uint32_t begin = blockIdx.x * chunk_a, end = min(begin + chunk_a, nitems);
  uint32_t i0 = begin + tid;
  uint32_t cnt = (i0 < end) ? min((end - i0 + BLOCK - 1u) / BLOCK, (uint32_t)kItems) : 0u;
  // Everything below sits in one loop over the schedule with the sweeps fully unrolled: left alone, the
  // compiler hoists every address that depends only on (thread, item, row) out of that loop -- a few
  // hundred values, spilled -- so the three values they derive from are made opaque per use.
  auto fresh = [&]() { asm volatile("" : "+v"(tid), "+v"(i0), "+v"(cnt)); };
  auto item_or_last = [&](uint32_t t) { return cnt ? i0 + min(t, cnt - 1u) * BLOCK : min(i0, nitems - 1u); };
  // items any thread of this workgroup owns (uniform).  PARTIAL (the host picks it when a workgroup's chunk leaves
  // whole items unused: shards well below 1M individuals): the item bodies nobody needs are skipped; the branches
  // cost the full-size kernel 5 %, so it runs without them -- an unused item is then processed as "missing".
  const uint32_t cnt_wg = !PARTIAL ? (uint32_t)kItems : begin < end ? min((end - begin + BLOCK - 1u) / BLOCK, (uint32_t)kItems) : 0u;
  const uint32_t g = blockIdx.x % (uint32_t)kResGroups, m = blockIdx.x / (uint32_t)kResGroups;

  if (n_sched == 0u) {
    if (blockIdx.x == 0) carry_state(S, W, J);
    return;
  }
  // the shard's weights: loaded once (two items in flight at a time), kept for the whole launch
  WT buf[kItems][KT];
#pragma unroll
  for (int t = 0; t < kItems; ++t) {
#pragma unroll
    for (int k = 0; k < KT; ++k) buf[t][k] = reinterpret_cast<const WT *>(w_a + (size_t)k * np)[item_or_last((uint32_t)t)];
    if (t & 1) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int t = 0; t < kItems; ++t)
    if (is_lds(t)) {
      const uint32_t i = item_or_last((uint32_t)t);
#pragma unroll
      for (int k = 0; k < KT; ++k) s_gam[lds_slot(t)][k][tid] = reinterpret_cast<const WT *>(p.gam + (size_t)k * np)[i];
      s_cn[lds_slot(t)][tid] = reinterpret_cast<const uint2 *>(p.cnt)[i];
    }
  auto get_item = [&](int t, WT (&wv)[KT]) {
#pragma unroll
    for (int k = 0; k < KT; ++k) wv[k] = buf[t][k];
  };
  auto put_item = [&](int t, const WT (&wv)[KT]) {
#pragma unroll
    for (int k = 0; k < KT; ++k) buf[t][k] = wv[k];
  };
  // the previous call's last SNP: its gamma step may be pending (column bits, exp(Elogbeta) of its last
  // pass), and its final values serve a first SNP at the same location
  uint32_t pword[kItems];
  {
    const uint32_t *pcol = reinterpret_cast<const uint32_t *>(p.bed + (size_t)sloc * p.colstride);
#pragma unroll
    for (int t = 0; t < kItems; ++t) pword[t] = svalid ? pcol[item_or_last((uint32_t)t) / kItemsPerWord] : 0x55555555u;
  }
  if (tid < J) {
    s_sb[tid] = S->eb[tid];
    s_plam[tid] = svalid ? p.lam[(size_t)sloc * J + tid] : 0.0;
    s_peb[tid] = svalid ? p.eb[(size_t)sloc * J + tid] : 0.0;
  }
  bool do_gamma = svalid != 0u && shol == 0u;
  bool prev_valid = svalid != 0u;
  uint32_t prev_loc = sloc, prev_hol = shol, prev_iters = siters;
  bool w_dirty = false;
  uint32_t xcount = 0u;  // exchanges of this launch
#ifdef TSAMD_SCHED_RAMP
  unsigned long long ramp_mark = wall_clock64();
  uint32_t ramp_idx = 0u;
#endif
#ifdef TSAMD_SCHED_TIME  // diagnostic build (tools/variant.sh): where a SNP's time goes, 10 ns ticks, workgroup 0
  unsigned long long tk_gamma = 0, tk_first = 0, tk_rest = 0, tk_xchg = 0, tk_head = 0, tk_tail = 0, tk_mark = wall_clock64();
  unsigned long long tk_fold = 0, tk_epi = 0;
  const unsigned long long tk_start = tk_mark;
#define TSAMD_TK(acc)                         \
  do {                                        \
    const unsigned long long now_ = wall_clock64(); \
    acc += now_ - tk_mark;                    \
    tk_mark = now_;                           \
  } while (0)
#else
#define TSAMD_TK(acc) \
  do {                \
  } while (0)
#endif
  __syncthreads();

  uint32_t iters = 0u;
  double lam_old = 0.0, eb_used = 0.0;
  double b0[KT], b1[KT], acc0[KT], acc1[KT];
  bool complete = false;
  auto begin_pass = [&]() {
    fresh();
    iters += 1u;
    lam_old = s_lam[tid < J ? tid : 0u];
    eb_used = s_eb[tid < J ? tid : 0u];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      b0[k] = uniform_f64(s_eb[2 * k]);
      b1[k] = uniform_f64(s_eb[2 * k + 1]);
      acc0[k] = acc1[k] = 0.0;
    }
  };
  auto consume = [&](uint32_t i, const WT (&wv)[KT], uint32_t wd) {
    const uint32_t code = wd >> (kCodeBits * (i % kItemsPerWord));
    double c0[2], c1[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      double mom, dad;
      bool ok;
      code_weights((code >> (2 * v)) & 3u, mom, dad, ok);
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const double wk = v ? wv[k].y : wv[k].x;
        s0 = fma(wk, b0[k], s0);
        s1 = fma(wk, b1[k], s1);
      }
      c0[v] = mom * fast_rcp(s0);
      c1[v] = dad * fast_rcp(s1);
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      acc0[k] = fma(c0[1], wv[k].y, acc0[k]);
      acc1[k] = fma(c1[1], wv[k].y, acc1[k]);
      acc0[k] = fma(c0[0], wv[k].x, acc0[k]);
      acc1[k] = fma(c1[0], wv[k].x, acc1[k]);
    }
  };
  // end of a pass: workgroup reduction, in-launch exchange, epilogue.  false: the exchange gave up.
  auto finish_pass = [&]() -> bool {
    fresh();
    const uint32_t lane = tid & 63u, wave = tid >> 6;
#ifdef TSAMD_SCHED_TIME
    const unsigned long long tf0 = wall_clock64();
#endif
    {
      using Fold = WaveFold<2 * KT>;
      double v[Fold::P];
#pragma unroll
      for (int q = 0; q < Fold::P; ++q) v[q] = 0.0;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        v[2 * k] = acc0[k];
        v[2 * k + 1] = acc1[k];
      }
      const double tot = Fold::fold(v, lane);
      const int slot = Fold::slot(lane);
      constexpr uint32_t kRep = 64 / Fold::P;
      if ((lane & (kRep - 1u)) == 0u && slot < (int)J) s_red[wave][slot] = tot;
    }
    __syncthreads();
#ifdef TSAMD_SCHED_TIME
    const unsigned long long tx0 = wall_clock64();
    tk_fold += tx0 - tf0;
#endif
    xcount += 1u;
    const uint32_t tag = xseq0 + xcount;
    if (tid < J) {
      double row = s_red[0][tid];
#pragma unroll
      for (int wv = 1; wv < kWaves; ++wv) row += s_red[wv][tid];
      const unsigned long long bits = __double_as_longlong(row);
      __hip_atomic_store(&xb->rows[g][m][2 * tid], ((unsigned long long)tag << 32) | (uint32_t)bits, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&xb->rows[g][m][2 * tid + 1], ((unsigned long long)tag << 32) | (uint32_t)(bits >> 32), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave == 0) {
      bool alive = true;
      if (m == 0) {
        unsigned v[kResMembers / 2];
        alive = res_sweep<kResMembers / 2>(&xb->rows[g][0][0], tag, 2 * J, g, gridDim.x, true, v, &xb->abort_word, p.host_error, lane);
        const double s = res_sum<kResMembers / 2>(v, lane);
        if (lane < 2 * J && !(lane & 1u)) {
          const unsigned long long bits = __double_as_longlong(s);
          const unsigned long long glo = ((unsigned long long)tag << 32) | (uint32_t)bits;
          const unsigned long long ghi = ((unsigned long long)tag << 32) | (uint32_t)(bits >> 32);
          if constexpr (WR == 0) {
            __hip_atomic_store(&xb->sums[tag & 1u][g][lane], glo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&xb->sums[tag & 1u][g][lane + 1], ghi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {  // sharded: the group sum goes to every rank (this one included), straight over xGMI
            for (uint32_t r = 0; r < p.xchg_world; ++r) {
              unsigned long long *dst = &p.peers[r]->res_sums[tag & 1u][p.xchg_rank * (uint32_t)kResGroups + g][lane];
              __hip_atomic_store(dst, glo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              __hip_atomic_store(dst + 1, ghi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
          }
        }
      }
      double s = 0.0;
      if constexpr (WR == 0) {
        unsigned v2[kResGroups / 2];
        alive = res_sweep<kResGroups / 2>(&xb->sums[tag & 1u][0][0], tag, 2 * J, 0u, gridDim.x, false, v2, &xb->abort_word,
                                          p.host_error, lane) && alive;
        s = res_sum<kResGroups / 2>(v2, lane);
      } else {  // (every rank runs at least 8 workgroups -- the host checks -- so all world * 8 rows exist)
        alive = res_sweep_ranks<WR>(&p.xchg->res_sums[tag & 1u][0][0], tag, 2 * J, p.xchg_world * (uint32_t)kResGroups, s,
                                    &xb->abort_word, p.host_error, lane) && alive;
      }
      if (lane < 2 * J && !(lane & 1u)) s_tot[lane >> 1] = s;
      if (lane == 0) s_alive = alive ? 1 : 0;
    }
    __syncthreads();
#ifdef TSAMD_SCHED_TIME
    const unsigned long long te0 = wall_clock64();
    tk_xchg += te0 - tx0;
#endif
    if (!s_alive) return false;
    if (tid < J) epilogue_values(p, s_tot[tid], eb_used, lam_old, s_lam, s_eb, s_diff);
    __syncthreads();
    complete = epilogue_complete(p, iters, J, s_diff);
#ifdef TSAMD_SCHED_TIME
    tk_epi += wall_clock64() - te0;
#endif
    return true;
  };

  // The next SNP's entry and its location's lambda / exp(Elogbeta) are requested one SNP ahead (after the
  // current SNP's first exchange: workgroup 0 has then published every earlier SNP of this launch with
  // agent-scope stores); a SNP at the location of its predecessor takes that one's final values from LDS.
  uint32_t loc = 0, hol = 0;
  uint32_t ent_n = sched[0];
  double nlam = 0.0, neb = 0.0;
  if (tid < J) {
    nlam = p.lam[(size_t)(ent_n & 0x7fffffffu) * J + tid];
    neb = p.eb[(size_t)(ent_n & 0x7fffffffu) * J + tid];
  }
  for (uint32_t idx = 0; idx < n_sched; ++idx) {
    const uint32_t ent = ent_n;
    loc = ent & 0x7fffffffu;
    hol = ent >> 31;
    ent_n = sched[min(idx + 1u, n_sched - 1u)];
    if (tid < J) {
      const bool local = prev_valid && loc == prev_loc;
      s_lam[tid] = local ? s_plam[tid] : nlam;
      s_eb[tid] = local ? s_peb[tid] : neb;
    }
    fresh();
    uint32_t word[kItems];
    {
      const uint32_t *col = reinterpret_cast<const uint32_t *>(p.bed + (size_t)loc * p.colstride);
#pragma unroll
      for (int t = 0; t < kItems; ++t) word[t] = col[item_or_last((uint32_t)t) / kItemsPerWord];
    }
    __syncthreads();
    iters = 0u;
    TSAMD_TK(tk_head);
    // ---- the previous SNP's gamma step (phi from the resident weights and the exp(Elogbeta) of that
    // SNP's last pass, read from LDS at each use).  Straight-line per item: an item past the end of the
    // thread's range is processed with "missing" codes and only its stores are guarded; an unobserved
    // genotype computes and discards (selects, no branch around assignments to the resident weights).
    if (do_gamma) {
      WT gs[KT];  // the streamed item in flight (requested one streamed item ahead)
      uint2 cs;
      auto load_gamma = [&](uint32_t i, WT (&gq)[KT], uint2 &cq) {
#pragma unroll
        for (int k = 0; k < KT; ++k) gq[k] = reinterpret_cast<const WT *>(p.gam + (size_t)k * np)[i];
        cq = reinterpret_cast<const uint2 *>(p.cnt)[i];
      };
      constexpr int kFirstStreamed = sched_next_streamed(-1, KT, kItems);
      if (kFirstStreamed < kItems) load_gamma(item_or_last((uint32_t)kFirstStreamed), gs, cs);
#pragma unroll
      for (int t = 0; t < kItems; ++t) {
        if (PARTIAL && (uint32_t)t >= cnt_wg) continue;
        fresh();
        const uint32_t i = item_or_last((uint32_t)t);
        const bool mine = (uint32_t)t < cnt;
        WT gv[KT];
        uint2 cv;
        if (is_lds(t)) {
#pragma unroll
          for (int k = 0; k < KT; ++k) gv[k] = s_gam[lds_slot(t)][k][tid];
          cv = s_cn[lds_slot(t)][tid];
        } else {
#pragma unroll
          for (int k = 0; k < KT; ++k) gv[k] = gs[k];
          cv = cs;
          constexpr int kNone = kItems;
          const int nxt = sched_next_streamed(t, KT, kItems);
          if (nxt < kNone) load_gamma(item_or_last((uint32_t)nxt), gs, cs);
        }
        __builtin_amdgcn_sched_barrier(0);
        WT wcur[KT];
        get_item(t, wcur);
        uint32_t pcode = mine ? pword[t] >> (kCodeBits * (i % kItemsPerWord)) : 0x5u;
        // the item's two individuals, one after the other through ONE copy of the code (a rolled loop that
        // works on the .x halves and swaps the halves after each turn: eight items times two individuals of
        // straight-line digamma / exp code would not fit the instruction cache)
#pragma unroll 1
        for (int v = 0; v < 2; ++v) {
          double mom, dad;
          bool ok;
          code_weights(pcode & 3u, mom, dad, ok);
          pcode >>= 2;
          double gx[KT], wx[KT];
          double s0 = 0.0, s1 = 0.0;
          uint32_t zo = 0u;  // (opaque zero: exp(Elogbeta) is re-read from LDS where it is used, not held in 32 registers)
          asm volatile("" : "+v"(zo));
          const double *sbv = s_sb + zo;
#pragma unroll
          for (int k = 0; k < KT; ++k) {
            gx[k] = gv[k].x;
            wx[k] = wcur[k].x;
            s0 = fma(wx[k], sbv[2 * k], s0);
            s1 = fma(wx[k], sbv[2 * k + 1], s1);
          }
          // update_gamma + update_rho_indiv (src/snpsamplinge.cc:688-719), as gamma_step_one with nodekappa = 0.5
          // (the host selects this kernel only then).  An unobserved genotype takes the same instructions with
          // a step size of exactly 0: gamma keeps its bits (its update term is finite), the weights are
          // recomputed from the unchanged gamma, c_n does not count -- no select per value.
          const double rho = ok ? fast_rsqrt(p.nodetau0 + (double)cv.x) : 0.0;
          const double c0 = mom * fast_rcp(s0), c1 = dad * fast_rcp(s1);
#pragma unroll
          for (int k = 0; k < KT; ++k) {
            const double e = c0 * (wx[k] * sbv[2 * k]) + c1 * (wx[k] * sbv[2 * k + 1]);
            gx[k] += rho * (p.alpha + p.gamma_scale * e - gx[k]);
          }
          gamma_to_w<KT>(gx, wx);
          const uint32_t cnew = ok ? cv.x + 1u : cv.x;
          cv.x = cv.y;
          cv.y = cnew;
#pragma unroll
          for (int k = 0; k < KT; ++k) {
            gv[k].x = gv[k].y;
            gv[k].y = gx[k];
            wcur[k].x = wcur[k].y;
            wcur[k].y = wx[k];
          }
        }
        if (is_lds(t)) {
#pragma unroll
          for (int k = 0; k < KT; ++k) s_gam[lds_slot(t)][k][tid] = gv[k];
          s_cn[lds_slot(t)][tid] = cv;
        } else if (mine) {
#pragma unroll
          for (int k = 0; k < KT; ++k) reinterpret_cast<WT *>(p.gam + (size_t)k * np)[i] = gv[k];
          reinterpret_cast<uint2 *>(p.cnt)[i] = cv;
        }
        put_item(t, wcur);
        __builtin_amdgcn_sched_barrier(0);
      }
      w_dirty = true;
    }
    TSAMD_TK(tk_gamma);
    // ---- first pass of the new SNP, from the resident weights like every later one --------------------
    begin_pass();
#pragma unroll
    for (int t = 0; t < kItems; ++t) {
      if (PARTIAL && (uint32_t)t >= cnt_wg) continue;
      fresh();
      WT wcur[KT];
      get_item(t, wcur);
      consume(item_or_last((uint32_t)t), wcur, (uint32_t)t < cnt ? word[t] : 0x55555555u);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!finish_pass()) return;
    TSAMD_TK(tk_first);
    fresh();
    if (tid < J) {
      nlam = __hip_atomic_load(&p.lam[(size_t)(ent_n & 0x7fffffffu) * J + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      neb = __hip_atomic_load(&p.eb[(size_t)(ent_n & 0x7fffffffu) * J + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    while (!complete) {
      begin_pass();
#pragma unroll
      for (int t = 0; t < kItems; ++t) {
        if (PARTIAL && (uint32_t)t >= cnt_wg) continue;
        fresh();
        WT wcur[KT];
        get_item(t, wcur);
        consume(item_or_last((uint32_t)t), wcur, (uint32_t)t < cnt ? word[t] : 0x55555555u);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!finish_pass()) return;
    }
    TSAMD_TK(tk_rest);
    // ---- the SNP is complete: s_lam / s_eb hold its final values, eb_used the exp(Elogbeta) its last
    // pass used.  Workgroup 0 publishes; everybody keeps what the next SNP's gamma step needs.
    if (blockIdx.x == 0) {
      if (tid < J) {
        __hip_atomic_store(&p.lam[(size_t)loc * J + tid], s_lam[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&p.eb[(size_t)loc * J + tid], s_eb[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (tid == 0) {
        count_snp(p, ctl, iters);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // published before this workgroup joins the next exchange
    }
    __syncthreads();
    if (tid < J) {
      s_sb[tid] = eb_used;
      s_plam[tid] = s_lam[tid];
      s_peb[tid] = s_eb[tid];
    }
#pragma unroll
    for (int t = 0; t < kItems; ++t) pword[t] = word[t];
    do_gamma = hol == 0u;
    prev_valid = true;
    prev_loc = loc;
    prev_hol = hol;
    prev_iters = iters;
    __syncthreads();
    TSAMD_TK(tk_tail);
#ifdef TSAMD_SCHED_RAMP  // diagnostic: time per SNP over ranges of the launch (does a launch start slow?)
    if (blockIdx.x == 0 && tid == 0) {
      const uint32_t marks[8] = {5u, 20u, 50u, 100u, 200u, 500u, 1000u, 2000u};
      for (int q = 0; q < 8; ++q)
        if (idx + 1u == marks[q]) {
          const unsigned long long now = wall_clock64();
          printf("ts_schedule ramp: SNPs up to %u: %.2f us per SNP in this range (launch of %u)\n", marks[q],
                 (now - ramp_mark) * 0.01 / (double)(marks[q] - ramp_idx), n_sched);
          ramp_mark = now;
          ramp_idx = marks[q];
        }
    }
#endif
  }

  // ---- end of the launch: the weights go back to memory, the state to the next call -------------
  if (w_dirty) {
#pragma unroll
    for (int t = 0; t < kItems; ++t) {
      fresh();
      if ((uint32_t)t < cnt) {
        const uint32_t i = i0 + (uint32_t)t * BLOCK;
        WT wcur[KT];
        get_item(t, wcur);
#pragma unroll
        for (int k = 0; k < KT; ++k) reinterpret_cast<WT *>(w_a + (size_t)k * np)[i] = wcur[k];
        if (is_lds(t)) {
#pragma unroll
          for (int k = 0; k < KT; ++k) reinterpret_cast<WT *>(p.gam + (size_t)k * np)[i] = s_gam[lds_slot(t)][k][tid];
          reinterpret_cast<uint2 *>(p.cnt)[i] = s_cn[lds_slot(t)][tid];
        }
      }
    }
  }
  fresh();
  if (blockIdx.x == 0) {
    if (tid < J) {
      W->lam[tid] = s_plam[tid];
      W->eb[tid] = s_sb[tid];
    }
    if (tid == 0) {
      W->idx = sidx + n_sched;
      W->valid = 1u;
      W->loc = prev_loc;
      W->hol = prev_hol;
      W->iters = prev_iters;
      W->done = 1u;
      W->nrows = 0u;
      W->epoch = epoch_now;
      ctl->xseq = xseq0 + xcount;
#ifdef TSAMD_SCHED_TIME
      if (n_sched >= 16u)
        printf("ts_schedule n=%u exchanges=%u | per SNP (us): head %.2f gamma %.2f first pass %.2f later passes %.2f tail %.2f | "
               "in exchanges %.2f, in folds %.2f, in epilogues %.2f | whole launch %.1f us\n", n_sched, xcount, tk_head * 0.01 / n_sched, tk_gamma * 0.01 / n_sched,
               tk_first * 0.01 / n_sched, tk_rest * 0.01 / n_sched, tk_tail * 0.01 / n_sched, tk_xchg * 0.01 / n_sched, tk_fold * 0.01 / n_sched, tk_epi * 0.01 / n_sched,
               (wall_clock64() - tk_start) * 0.01);
#endif
    }
  }
#undef TSAMD_TK
}

#ifdef TSAMD_MAIN_TU  // K-independent kernels: compiled into tsamd.hip only
// End of a schedule: complete the pending pass so that lambda/eb in the global arrays are
// final (whole SNPs only are ever enqueued, so the pending pass is the SNP's last).
// BLOCK is the workgroup size of the first-pass kernel, which would otherwise finish this pass
// (at the start of the next SNP): same (row group, value) mapping, same summation order, so a
// schedule cut anywhere gives the same bits as the uncut one.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void ts_flush(DevParams p, uint32_t par) {
  __shared__ double s_fin[BLOCK];
  __shared__ double s_lam[2 * TSAMD_MAX_K];
  __shared__ double s_eb[2 * TSAMD_MAX_K];
  __shared__ double s_diff[2 * TSAMD_MAX_K];
  Ctl *ctl = p.ctl;
  const State *S = &ctl->st[par ^ 1u];
  State *W = &ctl->st[par];
  const uint32_t J = 2 * p.K;
  const PendingIn pin = load_pending(S, J);
  if (p.xchg_world) publish_progress(p, S->epoch + 1ull);
  if (S->valid != 0u && S->done == 0u) {
    const uint32_t nrowsR = p.xchg_world ? p.xchg_world * S->nrows : p.rows_from_lt ? 1u : S->nrows;
    if (p.xchg_world) {
      wait_peer_rows(p, par ^ 1u, S->epoch, nrowsR);
      xchg_test_stall(p);
    }
    const double *rowsR = p.xchg_world  ? p.xchg->rows[par ^ 1u]
                          : p.rows_from_lt ? ctl->lt_sum[par ^ 1u]
                                           : p.partials + (size_t)(par ^ 1u) * kMaxGrid * J;
    const double vrow = p.xchg_world ? row_partial_sum_xchg<BLOCK>(rowsR, nrowsR, J) : row_partial_sum<BLOCK>(rowsR, nrowsR, J);
    finish_pending<BLOCK>(p, pin, vrow, J, s_fin, s_lam, s_eb, s_diff);
    publish_complete(p, ctl, S, W, J, s_lam, s_eb, true);
  } else {
    carry_state(S, W, J);
  }
}

// Start of a schedule of n entries.  drop_pending: forget the pending gamma step
// (tsamd_clear_pending).  n == 0xffffffff keeps the current schedule length.
__global__ void ts_begin(DevParams p, const uint32_t *host_sched, uint32_t *sched, uint32_t n, uint32_t par, uint32_t drop_pending) {
  Ctl *ctl = p.ctl;
  const uint32_t J = 2 * p.K;
  const State *S = &ctl->st[par ^ 1u];
  State *W = &ctl->st[par];
  if (p.xchg_world) publish_progress(p, S->epoch + 1ull);
  // (host_sched != NULL: take the entries straight from a pinned host buffer instead of a copy
  // enqueued ahead of this kernel -- measured slower for short schedules, not used)
  if (n != 0xffffffffu && host_sched)
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) sched[i] = host_sched[i];
  carry_state(S, W, J);
  __syncthreads();
  if (threadIdx.x == 0) {
    ctl->nxt[0].for_idx = ctl->nxt[1].for_idx = 0xffffffffu;  // captured for the previous schedule
    if (n != 0xffffffffu) {
      ctl->sched = sched;
      ctl->sched_len = n;
      W->idx = 0xffffffffu;
    }
    if (drop_pending) W->valid = 0u;
  }
}

// Sharded: fixed-order sum of the partial rows the pass of this parity wrote -> ctl->lt
__global__ __launch_bounds__(256) void ts_reduce_rows(DevParams p, uint32_t par) {
  __shared__ double s_fin[256];
  Ctl *ctl = p.ctl;
  const State *W = &ctl->st[par];
  const uint32_t tid = threadIdx.x, J = 2 * p.K;
  const uint32_t nrows = (W->valid != 0u && W->done == 0u) ? W->nrows : 0u;
  const double *rows = p.partials + (size_t)par * kMaxGrid * J;
  const uint32_t R = 256u / J;
  const double v = row_partial_sum<256>(rows, nrows, J);
  s_fin[tid] = v;
  __syncthreads();
  if (tid < J) {
    double lt = 0.0;
    for (uint32_t rr = 0; rr < R; ++rr) lt += s_fin[rr * J + tid];
    ctl->lt[par][tid] = lt;
  }
}

#endif  // TSAMD_MAIN_TU

template <int KT>
__global__ __launch_bounds__(kBlock) void ts_refresh_w(DevParams p) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= p.npairs) return;
  double ga[KT], gb[KT], wa[KT], wb[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const double2 v = reinterpret_cast<const double2 *>(p.gam + (size_t)k * p.npad)[i];
    ga[k] = v.x;
    gb[k] = v.y;
  }
  gamma_to_w<KT>(ga, wa);
  gamma_to_w<KT>(gb, wb);
#pragma unroll
  for (int k = 0; k < KT; ++k)
    reinterpret_cast<double2 *>(p.w + (size_t)k * p.npad)[i] = make_double2(wa[k], wb[k]);
}

// Host-side launcher of the K-specialised kernels; one translation unit per K
// (tsamd_inst.hip compiled with -DTSAMD_K=<k>) defines tsamd::launch_k<k>.
enum LaunchWhich { kLaunchPass = 0, kLaunchFirst = 1, kLaunchRefresh = 2, kLaunchResident = 3 };
using LaunchFn = void (*)(int which, uint32_t grid, uint32_t block, hipStream_t stream, const DevParams &p,
                          uint32_t par, uint32_t nrows_hint);

}  // namespace tsamd
