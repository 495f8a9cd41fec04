// HIP kernels of the SNP-minibatch SVI engine (gfx950).  See tsamd_device.h for
// the formulation and DESIGN.md for layouts and rooflines.
//
// Kernels (all stream-ordered, no host round trip between them):
//   ts_pass<KT, FIRST>   one inner pass over the shard's individuals for the current
//                        SNP: phi for both parents + lambda_t accumulation, deterministic
//                        block/grid reduction, and (single GPU) the K x 2 epilogue run by
//                        the last-arriving workgroup.  FIRST also applies the pending
//                        gamma/Elogtheta step of the previous SNP in the same sweep.
//   ts_epilogue<FIRST>   the K x 2 epilogue alone (multi-GPU: after the all-reduce).
//   ts_refresh_w<KT>     w = exp(psi(gamma) - rowmax) after tsamd_set_gamma.
//   ts_export_indiv      gamma / theta / Elogtheta as row-major [n][K].
//   ts_export_loc        Ebeta / Elogbeta from lambda.
//   ts_heldout_*         validation-mask fold and held-out log-likelihood terms.
//   ts_synth             synthetic PSD genotypes written straight into HBM.
#pragma once
#include "tsamd_device.h"

namespace tsamd {

// ---------------------------------------------------------------------------
// K x 2 epilogue of one pass: update_lambda + estimate_beta + convergence test
// (src/snpsamplinge.cc:356-364, :267-296; abs_mean src/matrix.hh:885-893).
// Called by every thread of ONE workgroup; thread j < 2K owns lambda[loc][j/2][j%2].
// lt = this thread's all-individual lambda_t[j];  ebj = exp(Elogbeta) this pass used.
template <bool FIRST>
__device__ __forceinline__ void epilogue_block(const DevParams &p, Ctl *ctl, uint32_t cur, uint32_t loc,
                                               uint32_t hol, double lt, double ebj, double *s_lam,
                                               double *s_diff) {
  const uint32_t tid = threadIdx.x;
  const uint32_t J = 2 * p.K;
  double nw = 0.0;
  if (tid < J) {
    double *lam = p.lam + (size_t)loc * J;
    const double old = lam[tid];
    nw = ((tid & 1u) ? p.eta1 : p.eta0) + lt;
    lam[tid] = nw;
    s_lam[tid] = nw;
    s_diff[tid] = fabs(nw - old);
  }
  __syncthreads();
  if (tid < J) {
    const double s = s_lam[tid & ~1u] + s_lam[tid | 1u];
    const double el = digamma(nw) - digamma(s);
    ctl->eb_stale[tid] = ebj;
    p.eb[(size_t)loc * J + tid] = exp(el);
  }
  if (tid == 0) {
    double d = 0.0;
    for (uint32_t j = 0; j < J; ++j) d += s_diff[j];
    d /= (double)J;
    const uint32_t it = FIRST ? 1u : ctl->iters + 1u;
    const uint32_t conv = (d < p.thresh) ? 1u : 0u;
    ctl->iters = it;
    ctl->done = (conv || it >= p.max_inner) ? 1u : 0u;
    ctl->last_iters = it;
    ctl->total_passes += 1ull;
    if (FIRST) {
      ctl->pend_loc = loc;
      ctl->pend_do = hol ? 0u : 1u;
      ctl->cursor = cur + 1u;
    }
  }
}

// w[k] = exp(psi(g[k]) - max_j psi(g[j])): Elogtheta up to a per-individual constant,
// which cancels in phi (estimate_theta, src/snpsamplinge.cc:721-740).
template <int KT>
__device__ __forceinline__ void gamma_to_w(const double (&g)[KT], double (&w)[KT]) {
  double ps[KT];
  double mx = -1.0e300;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    ps[k] = digamma(g[k]);
    mx = fmax(mx, ps[k]);
  }
#pragma unroll
  for (int k = 0; k < KT; ++k) w[k] = exp(ps[k] - mx);
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
  const double c0 = mom / s0, c1 = dad / s1;
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

// One inner pass for the current SNP.  KT == K exactly (one instantiation per K), so
// every k-loop is straight-line code and the K row loads of an iteration are issued
// back to back.  Item i of a workgroup's chunk is VEC consecutive individuals: one
// 8*VEC-byte load per population row and 2*VEC bits of the 2-bit column.  The plain
// pass uses VEC = 2; the first pass, which also carries the gamma step, uses VEC = 1 to
// halve its register footprint.
template <int KT, bool FIRST, int BLOCK, int VEC>
__global__ __launch_bounds__(BLOCK) void ts_pass(DevParams p) {
  constexpr int kWaves = BLOCK / 64;
  using LN = Lanes<VEC>;
  using WT = typename LN::T;
  constexpr uint32_t kItemsPerWord = 16u / VEC;  // items per 32-bit word of the column
  constexpr uint32_t kCodeBits = 2u * VEC;
  __shared__ double s_eb[2 * KT];
  __shared__ double s_sb[2 * KT];
  __shared__ double s_red[kWaves][2 * KT];
  __shared__ double s_fin[BLOCK];
  __shared__ double s_lam[2 * KT];
  __shared__ double s_diff[2 * KT];
  __shared__ uint32_t s_last;

  Ctl *ctl = p.ctl;
  const uint32_t tid = threadIdx.x;
  constexpr uint32_t J = 2 * KT;
  const uint32_t cur = ctl->cursor;
  const uint32_t idx = FIRST ? cur : cur - 1u;
  if (idx >= ctl->sched_len) return;
  if (!FIRST && ctl->done) return;
  const uint32_t ent = p.sched[idx];
  const uint32_t loc = ent & 0x7fffffffu, hol = ent >> 31;
  const bool do_gamma = FIRST && ctl->pend_do != 0u;
  const uint32_t prev_loc = ctl->pend_loc;

  if (tid < J) {
    s_eb[tid] = p.eb[(size_t)loc * J + tid];
    if (do_gamma) s_sb[tid] = ctl->eb_stale[tid];
  }
  __syncthreads();

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
  const uint32_t nitems = p.npad / VEC;
  const uint32_t chunk = FIRST ? p.chunk_first : p.chunk;
  const uint32_t begin = blockIdx.x * chunk;
  const uint32_t end = min(begin + chunk, nitems);
  const size_t np = p.npad;

  // ---- main sweep -------------------------------------------------------------------
  auto load_w = [&](uint32_t i, WT (&wv)[KT], uint32_t &word) {
#pragma unroll
    for (int k = 0; k < KT; ++k) wv[k] = reinterpret_cast<const WT *>(p.w + (size_t)k * np)[i];
    word = col[i / kItemsPerWord];  // lanes share addresses
  };
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
      c0[v] = mom / s0;
      c1[v] = dad / s1;
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
    // two-stage software pipeline: the next iteration's K row loads are in flight while
    // the current one is reduced
    auto consume = [&](uint32_t i, const WT (&wv)[KT], uint32_t word) {
      double w[VEC][KT];
      unpack_rows(wv, w);
      accumulate(i, w, word);
    };
    // (prefetch addresses are clamped, not predicated, so the load/wait counts are static)
    WT bufA[KT], bufB[KT];
    uint32_t wordA = 0, wordB = 0;
    uint32_t i = begin + tid;
    if (i < end) {
      load_w(i, bufA, wordA);
      while (true) {
        const uint32_t i1 = i + BLOCK;
        load_w(i1 < end ? i1 : i, bufB, wordB);
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the arithmetic
        consume(i, bufA, wordA);
        __builtin_amdgcn_sched_barrier(0);
        if (i1 >= end) break;
        const uint32_t i2 = i1 + BLOCK;
        load_w(i2 < end ? i2 : i1, bufA, wordA);
        __builtin_amdgcn_sched_barrier(0);
        consume(i1, bufB, wordB);
        __builtin_amdgcn_sched_barrier(0);
        if (i2 >= end) break;
        i = i2;
      }
    }
  } else {
    double sb0[KT], sb1[KT];  // exp(Elogbeta) of the previous SNP's last pass (wave-uniform)
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      sb0[k] = do_gamma ? uniform_f64(s_sb[2 * k]) : 0.0;
      sb1[k] = do_gamma ? uniform_f64(s_sb[2 * k + 1]) : 0.0;
    }
    for (uint32_t i = begin + tid; i < end; i += BLOCK) {
      WT wv[KT];
      uint32_t word;
      load_w(i, wv, word);
      double w[VEC][KT];
      if (do_gamma) {
        WT gv[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) gv[k] = reinterpret_cast<const WT *>(p.gam + (size_t)k * np)[i];
        typename LN::C cv = reinterpret_cast<const typename LN::C *>(p.cnt)[i];
        const uint32_t pcode = pcol[i / kItemsPerWord] >> (kCodeBits * (i % kItemsPerWord));
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
            gamma_step_one<KT>(g[v], w[v], sb0, sb1, mom, dad, cn[v], p);
            gamma_to_w<KT>(g[v], w[v]);
          }
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          double tg[VEC], tw[VEC];
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            tg[v] = g[v][k];
            tw[v] = w[v][k];
          }
          reinterpret_cast<WT *>(p.gam + (size_t)k * np)[i] = LN::pack(tg);
          reinterpret_cast<WT *>(p.w + (size_t)k * np)[i] = LN::pack(tw);
        }
        reinterpret_cast<typename LN::C *>(p.cnt)[i] = LN::pack_c(cn);
      } else {
        unpack_rows(wv, w);
      }
      accumulate(i, w, word);
    }
  }

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
  if (tid < J) {
    double v = s_red[0][tid];
#pragma unroll
    for (int wv = 1; wv < kWaves; ++wv) v += s_red[wv][tid];
    if (p.tail == 2u)
      p.partials[(size_t)blockIdx.x * J + tid] = v;
    else
      st_agent(p.partials + (size_t)blockIdx.x * J + tid, v);
  }
  if (p.tail == 2u) return;  // ts_finish (next kernel) adds the partial rows up
  // hand-off to the last-arriving workgroup.  The partial rows are written through to
  // the coherence point (agent-scope sc1 stores) and read back with agent-scope loads, so
  // no L2 write-back / L1 invalidate is needed: every storing wave drains its stores,
  // the workgroup barrier orders them before lane 0's arrival on the ticket, and the
  // last arriver reads the rows only after its ticket value has returned.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const uint32_t t = __hip_atomic_fetch_add(&ctl->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == gridDim.x - 1u) ? 1u : 0u;
  }
  __syncthreads();
  asm volatile("" ::: "memory");
  if (!s_last) return;

  // grid reduction by the last workgroup, fixed order: thread (r, j) sums
  // partials[g][j] for g = r, r+R, ... (loads batched 8 deep); then r = 0..R-1.
  constexpr uint32_t R = BLOCK / J;  // J <= 64 -> R >= 4
  const uint32_t j = tid % J, r = tid / J;
  const uint32_t G = gridDim.x;
  double v = 0.0;
  if (r < R) {
    for (uint32_t g0 = r; g0 < G; g0 += 8u * R) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t g = g0 + (uint32_t)u * R;
        t[u] = ld_agent(p.partials + (size_t)min(g, G - 1u) * J + j);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) v += (g0 + (uint32_t)u * R < G) ? t[u] : 0.0;
    }
  }
  s_fin[tid] = v;
  __syncthreads();
  double lt = 0.0;
  if (tid < J) {
    for (uint32_t rr = 0; rr < R; ++rr) lt += s_fin[rr * J + tid];
    lt *= s_eb[tid];  // the b[k,t] factored out of the accumulation
  }
  if (tid == 0) ctl->ticket = 0u;
  if (p.tail == 1u) {
    if (tid < J) ctl->lt[tid] = lt;
    return;
  }
  epilogue_block<FIRST>(p, ctl, cur, loc, hol, lt, (tid < J) ? s_eb[tid] : 0.0, s_lam, s_diff);
}

// tail == 2: the partial rows of the preceding ts_pass launch (grid G) are added up in
// the same fixed order by one workgroup after the kernel boundary; then either the
// epilogue (single GPU) or ctl->lt for the all-reduce (to_lt).
template <bool FIRST>
__global__ __launch_bounds__(256) void ts_finish(DevParams p, uint32_t G, uint32_t to_lt) {
  __shared__ double s_fin[256];
  __shared__ double s_lam[2 * TSAMD_MAX_K];
  __shared__ double s_diff[2 * TSAMD_MAX_K];
  Ctl *ctl = p.ctl;
  const uint32_t tid = threadIdx.x, J = 2 * p.K;
  const uint32_t cur = ctl->cursor;
  const uint32_t idx = FIRST ? cur : cur - 1u;
  if (idx >= ctl->sched_len) return;
  if (!FIRST && ctl->done) return;
  const uint32_t ent = p.sched[idx];
  const uint32_t loc = ent & 0x7fffffffu, hol = ent >> 31;
  const uint32_t R = 256u / J;
  const uint32_t j = tid % J, r = tid / J;
  double v = 0.0;
  if (r < R)
    for (uint32_t g0 = r; g0 < G; g0 += 8u * R) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = p.partials[(size_t)min(g0 + (uint32_t)u * R, G - 1u) * J + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += (g0 + (uint32_t)u * R < G) ? t[u] : 0.0;
    }
  s_fin[tid] = v;
  __syncthreads();
  const double ebj = (tid < J) ? p.eb[(size_t)loc * J + tid] : 0.0;
  double lt = 0.0;
  if (tid < J) {
    for (uint32_t rr = 0; rr < R; ++rr) lt += s_fin[rr * J + tid];
    lt *= ebj;
  }
  if (to_lt) {
    if (tid < J) ctl->lt[tid] = lt;
    return;
  }
  epilogue_block<FIRST>(p, ctl, cur, loc, hol, lt, ebj, s_lam, s_diff);
}

// multi-GPU: epilogue after the RCCL all-reduce of ctl->lt into ctl->lt_sum
template <bool FIRST>
__global__ __launch_bounds__(64) void ts_epilogue(DevParams p) {
  __shared__ double s_lam[2 * TSAMD_MAX_K];
  __shared__ double s_diff[2 * TSAMD_MAX_K];
  Ctl *ctl = p.ctl;
  const uint32_t tid = threadIdx.x, J = 2 * p.K;
  const uint32_t cur = ctl->cursor;
  const uint32_t idx = FIRST ? cur : cur - 1u;
  if (idx >= ctl->sched_len) return;
  if (!FIRST && ctl->done) return;
  const uint32_t ent = p.sched[idx];
  const uint32_t loc = ent & 0x7fffffffu, hol = ent >> 31;
  const double lt = (tid < J) ? ctl->lt_sum[tid] : 0.0;
  const double ebj = (tid < J) ? p.eb[(size_t)loc * J + tid] : 0.0;
  epilogue_block<FIRST>(p, ctl, cur, loc, hol, lt, ebj, s_lam, s_diff);
}

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
enum LaunchWhich { kLaunchPass = 0, kLaunchFirst = 1, kLaunchRefresh = 2 };
using LaunchFn = void (*)(int which, uint32_t grid, uint32_t block, hipStream_t stream, const DevParams &p);

}  // namespace tsamd
