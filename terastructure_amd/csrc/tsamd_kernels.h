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
__device__ __forceinline__ void gamma_to_w(const double (&g)[KT], double (&w)[KT], uint32_t K) {
  double ps[KT];
  double mx = -1.0e300;
#pragma unroll
  for (int k = 0; k < KT; ++k)
    if (k < (int)K) {
      ps[k] = digamma(g[k]);
      mx = fmax(mx, ps[k]);
    }
#pragma unroll
  for (int k = 0; k < KT; ++k)
    if (k < (int)K) w[k] = exp(ps[k] - mx);
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
  for (int k = 0; k < KT; ++k)
    if (k < (int)p.K) {
      s0 = fma(w[k], sb0[k], s0);
      s1 = fma(w[k], sb1[k], s1);
    }
  const double base = p.nodetau0 + (double)c;
  const double rho = (p.nodekappa == 0.5) ? 1.0 / sqrt(base) : pow(base, -p.nodekappa);
  c += 1u;
  const double c0 = mom / s0, c1 = dad / s1;
#pragma unroll
  for (int k = 0; k < KT; ++k)
    if (k < (int)p.K) {
      const double e = c0 * (w[k] * sb0[k]) + c1 * (w[k] * sb1[k]);  // y*phi_mom + (2-y)*phi_dad
      g[k] += rho * (p.alpha + p.gamma_scale * e - g[k]);
    }
}

template <int KT, bool FIRST>
__global__ __launch_bounds__(kBlock) void ts_pass(DevParams p) {
  __shared__ double s_eb[2 * KT];
  __shared__ double s_sb[2 * KT];
  __shared__ double s_red[kWaves][2 * KT];
  __shared__ double s_fin[kBlock];
  __shared__ double s_lam[2 * KT];
  __shared__ double s_diff[2 * KT];
  __shared__ uint32_t s_last;

  Ctl *ctl = p.ctl;
  const uint32_t tid = threadIdx.x;
  const uint32_t K = p.K, J = 2 * K;
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

  double b0[KT], b1[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    b0[k] = (k < (int)K) ? s_eb[2 * k] : 0.0;
    b1[k] = (k < (int)K) ? s_eb[2 * k + 1] : 0.0;
  }
  double acc0[KT], acc1[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) acc0[k] = acc1[k] = 0.0;

  const uint8_t *col = p.bed + (size_t)loc * p.colstride;
  const uint8_t *pcol = p.bed + (size_t)prev_loc * p.colstride;
  const uint32_t begin = blockIdx.x * p.chunk;
  const uint32_t end = min(begin + p.chunk, p.npairs);

  for (uint32_t i = begin + tid; i < end; i += kBlock) {
    double wa[KT], wb[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (k < (int)K) {
        const double2 v = reinterpret_cast<const double2 *>(p.w + (size_t)k * p.npad)[i];
        wa[k] = v.x;
        wb[k] = v.y;
      }
    const uint32_t code = ((uint32_t)col[i >> 1] >> (4u * (i & 1u))) & 0xfu;

    if (FIRST && do_gamma) {
      double sb0[KT], sb1[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        sb0[k] = (k < (int)K) ? s_sb[2 * k] : 0.0;
        sb1[k] = (k < (int)K) ? s_sb[2 * k + 1] : 0.0;
      }
      double ga[KT], gb[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < (int)K) {
          const double2 v = reinterpret_cast<const double2 *>(p.gam + (size_t)k * p.npad)[i];
          ga[k] = v.x;
          gb[k] = v.y;
        }
      uint2 cn = reinterpret_cast<const uint2 *>(p.cnt)[i];
      const uint32_t pcode = ((uint32_t)pcol[i >> 1] >> (4u * (i & 1u))) & 0xfu;
      double mom, dad;
      bool ok;
      code_weights(pcode & 3u, mom, dad, ok);
      if (ok) {
        gamma_step_one<KT>(ga, wa, sb0, sb1, mom, dad, cn.x, p);
        gamma_to_w<KT>(ga, wa, K);
      }
      code_weights(pcode >> 2, mom, dad, ok);
      if (ok) {
        gamma_step_one<KT>(gb, wb, sb0, sb1, mom, dad, cn.y, p);
        gamma_to_w<KT>(gb, wb, K);
      }
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < (int)K) {
          reinterpret_cast<double2 *>(p.gam + (size_t)k * p.npad)[i] = make_double2(ga[k], gb[k]);
          reinterpret_cast<double2 *>(p.w + (size_t)k * p.npad)[i] = make_double2(wa[k], wb[k]);
        }
      reinterpret_cast<uint2 *>(p.cnt)[i] = cn;
    }

    double ma, da, mb, db;
    bool oka, okb;
    code_weights(code & 3u, ma, da, oka);
    code_weights(code >> 2, mb, db, okb);
    double s0a = 0.0, s1a = 0.0, s0b = 0.0, s1b = 0.0;
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (k < (int)K) {
        s0a = fma(wa[k], b0[k], s0a);
        s1a = fma(wa[k], b1[k], s1a);
        s0b = fma(wb[k], b0[k], s0b);
        s1b = fma(wb[k], b1[k], s1b);
      }
    const double ca0 = ma / s0a, ca1 = da / s1a, cb0 = mb / s0b, cb1 = db / s1b;
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (k < (int)K) {
        acc0[k] = fma(ca0, wa[k], fma(cb0, wb[k], acc0[k]));
        acc1[k] = fma(ca1, wa[k], fma(cb1, wb[k], acc1[k]));
      }
  }

  // workgroup reduction, fixed order: lanes (xor tree) -> waves (0..3)
  const uint32_t lane = tid & 63u, wave = tid >> 6;
#pragma unroll
  for (int k = 0; k < KT; ++k)
    if (k < (int)K) {
      const double r0 = wave_sum(acc0[k]);
      const double r1 = wave_sum(acc1[k]);
      if (lane == 0) {
        s_red[wave][2 * k] = r0;
        s_red[wave][2 * k + 1] = r1;
      }
    }
  __syncthreads();
  if (tid < J) {
    double v = s_red[0][tid];
#pragma unroll
    for (int wv = 1; wv < kWaves; ++wv) v += s_red[wv][tid];
    st_agent(p.partials + (size_t)blockIdx.x * J + tid, v);
  }
  // hand-off to the last-arriving workgroup: drain stores, release, ticket, acquire
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t t = __hip_atomic_fetch_add(&ctl->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t last = (t == gridDim.x - 1u) ? 1u : 0u;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;

  // grid reduction by the last workgroup, fixed order: thread (r, j) sums
  // partials[g][j] for g = r, r+R, ...; then r = 0..R-1.
  const uint32_t R = kBlock / J;  // J <= 64 -> R >= 4
  const uint32_t j = tid % J, r = tid / J;
  double v = 0.0;
  if (r < R)
    for (uint32_t g = r; g < gridDim.x; g += R) v += ld_agent(p.partials + (size_t)g * J + j);
  s_fin[tid] = v;
  __syncthreads();
  double lt = 0.0;
  if (tid < J) {
    for (uint32_t rr = 0; rr < R; ++rr) lt += s_fin[rr * J + tid];
    lt *= s_eb[tid];  // the b[k,t] factored out of the accumulation
  }
  if (tid == 0) ctl->ticket = 0u;
  if (p.multi) {
    if (tid < J) ctl->lt[tid] = lt;
    return;
  }
  epilogue_block<FIRST>(p, ctl, cur, loc, hol, lt, (tid < J) ? s_eb[tid] : 0.0, s_lam, s_diff);
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
  for (int k = 0; k < KT; ++k)
    if (k < (int)p.K) {
      const double2 v = reinterpret_cast<const double2 *>(p.gam + (size_t)k * p.npad)[i];
      ga[k] = v.x;
      gb[k] = v.y;
    }
  gamma_to_w<KT>(ga, wa, p.K);
  gamma_to_w<KT>(gb, wb, p.K);
#pragma unroll
  for (int k = 0; k < KT; ++k)
    if (k < (int)p.K) reinterpret_cast<double2 *>(p.w + (size_t)k * p.npad)[i] = make_double2(wa[k], wb[k]);
}

// mode 0: gamma, 1: theta = gamma / sum, 2: Elogtheta = psi(gamma) - psi(sum)
// (estimate_all_theta src/snpsamplinge.cc:595-609, set_dir_exp src/lib.hh:19-35);
// out is row-major [n_out][K]; rows = list of local individual ids or NULL for 0..n_out-1.
__global__ void ts_export_indiv(const double *gam, uint32_t npad, uint32_t K, uint32_t n_out,
                                const uint32_t *rows, int mode, double *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const uint32_t n = rows ? rows[i] : i;
  double s = 0.0;
  for (uint32_t k = 0; k < K; ++k) s += gam[(size_t)k * npad + n];
  const double ps = (mode == 2) ? digamma(s) : 0.0;
  for (uint32_t k = 0; k < K; ++k) {
    const double g = gam[(size_t)k * npad + n];
    out[(size_t)i * K + k] = (mode == 0) ? g : (mode == 1) ? g / s : digamma(g) - ps;
  }
}

// mode 0: Ebeta[loc][k] = l0/(l0+l1); mode 1: Elogbeta[loc][k][t]; mode 2: exp(Elogbeta) into eb
// (estimate_beta, src/snpsamplinge.cc:279-296)
__global__ void ts_export_loc(const double *lam, uint32_t K, uint32_t first_loc, uint32_t n_locs, int mode,
                              double *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_locs * K) return;
  const uint32_t loc = first_loc + i / K, k = i % K;
  const double l0 = lam[((size_t)loc * K + k) * 2], l1 = lam[((size_t)loc * K + k) * 2 + 1];
  double s = 0.0;
  s += l0;
  s += l1;
  if (mode == 0) {
    out[i] = l0 / s;
  } else {
    const double ps = digamma(s);
    const double e0 = digamma(l0) - ps, e1 = digamma(l1) - ps;
    double *o = (mode == 1) ? out + (size_t)i * 2 : out + ((size_t)loc * K + k) * 2;
    o[0] = (mode == 1) ? e0 : exp(e0);
    o[1] = (mode == 1) ? e1 : exp(e1);
  }
}

// fold validation entries into the column as "missing" (01) and return the true codes
__global__ void ts_heldout_fold(uint8_t *col, const uint32_t *local_ids, uint32_t count, uint8_t *orig) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t n = local_ids[i];
  uint32_t *word = reinterpret_cast<uint32_t *>(col) + (n >> 4);
  const uint32_t sh = 2u * (n & 15u);
  const uint32_t old = atomicOr(word, 1u << sh);
  atomicAnd(word, ~(2u << sh));
  orig[i] = (uint8_t)((old >> sh) & 3u);
}

// per-entry held-out log-likelihood term (snp_likelihood, src/snpsamplinge.hh:336-360)
__global__ void ts_heldout_ll(const double *gam, uint32_t npad, uint32_t K, const double *lam_loc,
                              const uint32_t *local_ids, const uint8_t *ytrue, uint32_t count, double *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t n = local_ids[i];
  double s = 0.0;
  for (uint32_t k = 0; k < K; ++k) s += gam[(size_t)k * npad + n];
  double q = 0.0;
  for (uint32_t k = 0; k < K; ++k) {
    const double l0 = lam_loc[2 * k], l1 = lam_loc[2 * k + 1];
    double ls = 0.0;
    ls += l0;
    ls += l1;
    q += (l0 / ls) * (gam[(size_t)k * npad + n] / s);
  }
  const int x = ytrue[i];
  const double v = (x == 1) ? 2.0 : 1.0;  // 2!/(x!(2-x)!)
  double sum = v * pow(q, (double)x) * pow(1.0 - q, (double)(2 - x));
  if (sum < 1e-30) sum = 1e-30;
  out[i] = log(sum);
}

// ---------------------------------------------------------------------------
// Synthetic Pritchard-Stephens-Donnelly genotypes: one thread makes one
// column byte (4 individuals) for CT consecutive columns.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

constexpr int kSynthCols = 8;

__global__ __launch_bounds__(kBlock) void ts_synth(uint8_t *bed, uint64_t colstride, const double *theta_kmajor,
                                                  uint32_t npad, uint32_t n_local, uint32_t n_begin, uint32_t K,
                                                  const double *beta, uint32_t first_loc, uint32_t n_locs,
                                                  uint64_t seed, double missing_rate) {
  __shared__ double s_beta[kSynthCols * TSAMD_MAX_K];
  const uint32_t q = blockIdx.x * kBlock + threadIdx.x;  // quad of individuals
  const uint32_t c0 = blockIdx.y * kSynthCols;
  const uint32_t nc = min((uint32_t)kSynthCols, n_locs - c0);
  for (uint32_t t = threadIdx.x; t < nc * K; t += kBlock) s_beta[t] = beta[(size_t)c0 * K + t];
  __syncthreads();
  if (q >= npad / 4) return;
  double pr[kSynthCols][4];
#pragma unroll
  for (int c = 0; c < kSynthCols; ++c)
#pragma unroll
    for (int u = 0; u < 4; ++u) pr[c][u] = 0.0;
  for (uint32_t k = 0; k < K; ++k) {
    const double2 t01 = reinterpret_cast<const double2 *>(theta_kmajor + (size_t)k * npad)[2 * q];
    const double2 t23 = reinterpret_cast<const double2 *>(theta_kmajor + (size_t)k * npad)[2 * q + 1];
#pragma unroll
    for (int c = 0; c < kSynthCols; ++c) {
      const double b = (c < (int)nc) ? s_beta[c * K + k] : 0.0;
      pr[c][0] = fma(t01.x, b, pr[c][0]);
      pr[c][1] = fma(t01.y, b, pr[c][1]);
      pr[c][2] = fma(t23.x, b, pr[c][2]);
      pr[c][3] = fma(t23.y, b, pr[c][3]);
    }
  }
#pragma unroll
  for (int c = 0; c < kSynthCols; ++c) {
    if (c >= (int)nc) break;
    const uint32_t loc = first_loc + c0 + c;
    uint32_t byte = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t nl = 4 * q + u;
      uint32_t code = 1u;  // padding individuals are missing
      if (nl < n_local) {
        const uint64_t key = ((uint64_t)loc << 32) | (uint64_t)(n_begin + nl);
        const uint64_t h = mix64(mix64(seed) ^ key);
        const double u1 = (double)(uint32_t)(h >> 32) * (1.0 / 4294967296.0);
        const double u2 = (double)(uint32_t)h * (1.0 / 4294967296.0);
        const uint32_t y = (u1 < pr[c][u] ? 1u : 0u) + (u2 < pr[c][u] ? 1u : 0u);
        code = (y == 0u) ? 0u : (y == 1u) ? 2u : 3u;
        if (missing_rate > 0.0) {
          const uint64_t h2 = mix64(h);
          if ((double)(uint32_t)(h2 >> 32) * (1.0 / 4294967296.0) < missing_rate) code = 1u;
        }
      }
      byte |= code << (2 * u);
    }
    bed[(size_t)loc * colstride + q] = (uint8_t)byte;
  }
}

}  // namespace tsamd
