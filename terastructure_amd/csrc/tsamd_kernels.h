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
// The register-resident kernels (ts_resident, ts_schedule) live in tsamd_resident_kernels.h.
#pragma once
#include "tsamd_device.h"

namespace tsamd {

// An earlier resident launch of this context gave up (the first word of its exchange buffer, ResXchg::abort_word in
// tsamd_resident_kernels.h): every later kernel of the sequence returns without touching anything, so that the state the
// failed launch started from is still there when the host deals with it (tsamd_synchronize replays from it).
__device__ __forceinline__ bool sequence_aborted(const DevParams &p) {
  return p.res != nullptr &&
         __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p.res), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
}

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
// The K x 2 epilogue proper, for the thread that holds lt = the row total of value j (threads j and j ^ 1 are
// neighbouring lanes of one wave): lambda_t = eb_used * lt, update_lambda, estimate_beta; outputs into LDS.
__device__ __forceinline__ void epilogue_values_reg(const DevParams &p, uint32_t j, double lt, double eb_used, double lam_old,
                                                    double &nw, double &eb_new, double &diff) {
  // eta + b[k,t] * (row sum): the b factored out of the accumulation; an explicit fma so that
  // every kernel that inlines this rounds the same way whatever the compiler would contract
  nw = fma(lt, eb_used, (j & 1u) ? p.eta1 : p.eta0);
  // exp(Elogbeta_kt) = exp(psi(lambda_kt) - psi(lambda_k0 + lambda_k1)) without a log: both
  // digammas in the split form z * exp(a) (tsamd_device.h), side by side in one instruction
  // stream; the pair sum comes from the neighbouring lane (t = 0/1 are adjacent threads)
  const double pair = nw + partner<1>(nw);
  double z1, a1, z2, a2;
  exp_digamma_split(nw, z1, a1);
  exp_digamma_split(pair, z2, a2);
  eb_new = (z1 * fast_rcp(z2)) * exp_nonpos(a1 - a2);
  diff = fabs(nw - lam_old);
}
__device__ __forceinline__ void epilogue_values_at(const DevParams &p, uint32_t j, double lt, double eb_used, double lam_old,
                                                   double *s_lam, double *s_eb, double *s_diff) {
  double nw, eb_new, diff;
  epilogue_values_reg(p, j, lt, eb_used, lam_old, nw, eb_new, diff);
  s_lam[j] = nw;
  s_eb[j] = eb_new;
  s_diff[j] = diff;
}
// ... called by threads tid < J for value tid
__device__ __forceinline__ void epilogue_values(const DevParams &p, double lt, double eb_used, double lam_old,
                                                double *s_lam, double *s_eb, double *s_diff) {
  epilogue_values_at(p, threadIdx.x, lt, eb_used, lam_old, s_lam, s_eb, s_diff);
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

// The same decision taken by a whole wave (J <= 64), for wide rows: the sequential sum costs every thread 2 J dependent
// instructions per pass (K = 20: 80).  When the pass cap decides nothing is summed.  Otherwise lane j reads |dlambda_j| and
// the wave adds them in a butterfly (every lane gets the same bits); a tree rounds differently from the reference's
// sequential order in the last bits, so the sequential sum is taken -- by everybody, the branch is uniform -- exactly when the
// mean lies within 1e-9 (relative) of the threshold, where those bits could matter.  Always epilogue_complete's decision.
template <uint32_t J>
__device__ __forceinline__ bool epilogue_complete_wave(const DevParams &p, uint32_t iters, const double *s_diff, uint32_t lane) {
  static_assert(J <= 64u, "one value per lane");
  if (iters >= p.max_inner) return true;
  double v = lane < J ? s_diff[lane] : 0.0;
  v += partner<1>(v);
  v += partner<2>(v);
  v += partner<4>(v);
  v += partner<8>(v);
  v = pair_add<16>(v, v);
  v = pair_add<32>(v, v);
  const double mean = (J & (J - 1u)) == 0u ? v * (1.0 / (double)J) : v / (double)J;
  if (fabs(mean - p.thresh) > 1.0e-9 * p.thresh) return mean < p.thresh;
  return epilogue_complete(p, iters, J, s_diff);
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
  const bool aborted = sequence_aborted(p);
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
  if (aborted) return;
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
      // (the first pass is arithmetic-bound next to its gamma step: reciprocal estimate + one third-order
      // step there, IEEE division in the bandwidth-bound plain pass)
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
  if (sequence_aborted(p)) return;
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
  if (sequence_aborted(p)) return;
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
                          uint32_t par, uint32_t nrows_hint, uint32_t serial);

}  // namespace tsamd
