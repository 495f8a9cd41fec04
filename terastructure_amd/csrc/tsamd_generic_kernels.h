// Kernels that do not depend on K at compile time (accessors, validation mask,
// synthetic data).  Included by tsamd.hip only.
#pragma once
#include "tsamd_device.h"

namespace tsamd {

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

// after a direct (unstaged) upload of columns [first_loc, first_loc + n_locs): the individuals past
// the shard's end that share its last byte (neighbours' bits, or PLINK's zero padding) -> missing
__global__ void ts_fix_tail(uint8_t *bed, uint64_t colstride, uint32_t first_loc, uint32_t n_locs, uint64_t last_byte,
                            uint32_t keep_mask) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_locs) return;
  uint8_t *b = bed + (size_t)(first_loc + j) * colstride + last_byte;
  *b = (uint8_t)((*b & keep_mask) | (0x55u & ~keep_mask));
}

// PLINK individual-major payload -> the SNP-major 2-bit columns the engine keeps (a 2-bit matrix
// transpose).  rows: [n_rows][row_bytes] staged on the device, row r = individual (first local id +
// r), location j at bits 2(j%4) of byte j/4.  One workgroup transposes a tile of kTrRows
// individuals x 4*kTrBytes locations through LDS: coalesced row reads in, one 32-bit word (16
// individuals of one location) per store out, 64 contiguous bytes per location and tile.
// first_local must be a multiple of 16 (whole output words); rows past n_rows read as missing.
constexpr int kTrRows = 256;   // individuals per tile
constexpr int kTrBytes = 64;   // input bytes per row and tile (256 locations)
__global__ __launch_bounds__(256) void ts_transpose_indiv_major(const uint8_t *rows, uint64_t row_bytes, uint32_t n_rows,
                                                                uint32_t first_local, uint32_t first_loc, uint32_t n_locs,
                                                                uint8_t *bed, uint64_t colstride) {
  __shared__ uint8_t tile[kTrRows][kTrBytes + 4];  // (+4: row stride 68 bytes spreads the column reads over the banks)
  const uint32_t r0 = blockIdx.x * kTrRows, q0 = blockIdx.y * kTrBytes;
  const uint32_t nbytes = (n_locs + 3u) / 4u;  // payload bytes per row that carry locations [first_loc, first_loc + n_locs)
  for (uint32_t e = threadIdx.x; e < kTrRows * kTrBytes; e += 256u) {
    const uint32_t r = e / kTrBytes, q = e % kTrBytes;
    uint8_t v = 0x55;  // missing
    if (r0 + r < n_rows && q0 + q < nbytes) v = rows[(size_t)(r0 + r) * row_bytes + q0 + q];
    tile[r][q] = v;
  }
  __syncthreads();
  // thread -> (input byte position q, group of 64 individuals g): 4 locations x 4 output words
  const uint32_t q = threadIdx.x % kTrBytes, g = threadIdx.x / kTrBytes;  // g in 0..3
  uint32_t out[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int wd = 0; wd < 4; ++wd) out[j][wd] = 0u;
#pragma unroll
  for (int wd = 0; wd < 4; ++wd)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const uint32_t b = tile[g * 64u + wd * 16u + i][q];
#pragma unroll
      for (int j = 0; j < 4; ++j) out[j][wd] |= ((b >> (2 * j)) & 3u) << (2 * i);
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t l = 4u * (q0 + q) + j;
    if (l >= n_locs) continue;
    uint32_t *col = reinterpret_cast<uint32_t *>(bed + (size_t)(first_loc + l) * colstride) + (first_local + r0 + g * 64u) / 16u;
#pragma unroll
    for (int wd = 0; wd < 4; ++wd)
      if (r0 + g * 64u + wd * 16u < n_rows) col[wd] = out[j][wd];  // (a word's individuals past n_rows are written as missing)
  }
}

// counts of the four PLINK codes over the shard's real individuals in columns [first_loc, +n_locs)
// (SNP::read_bed's tallies for param.txt, src/snp.cc:203-216, :245-247); one workgroup per column
// slice, 64-bit words, three popcounts per word; out[4] accumulated with atomics.
__global__ __launch_bounds__(256) void ts_count_codes(const uint8_t *bed, uint64_t colstride, uint32_t first_loc,
                                                      uint32_t n_local, unsigned long long *out) {
  __shared__ unsigned long long s_cnt[4];
  if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0ull;
  __syncthreads();
  const uint64_t *col = reinterpret_cast<const uint64_t *>(bed + (size_t)(first_loc + blockIdx.y) * colstride);
  const uint32_t nwords = (n_local + 31u) / 32u;  // 32 individuals per 64-bit word
  unsigned long long c01 = 0, c10 = 0, c11 = 0, tot = 0;
  for (uint32_t wi = blockIdx.x * 256u + threadIdx.x; wi < nwords; wi += gridDim.x * 256u) {
    uint64_t x = col[wi];
    const uint32_t valid = min(32u, n_local - wi * 32u);
    const uint64_t m = valid == 32u ? 0x5555555555555555ull : ((1ull << (2u * valid)) - 1ull) & 0x5555555555555555ull;
    const uint64_t lo = x & m, hi = (x >> 1) & m;
    c01 += __popcll(lo & ~hi);
    c10 += __popcll(hi & ~lo);
    c11 += __popcll(hi & lo);
    tot += valid;
  }
  atomicAdd(&s_cnt[0], tot - c01 - c10 - c11);
  atomicAdd(&s_cnt[1], c01);
  atomicAdd(&s_cnt[2], c10);
  atomicAdd(&s_cnt[3], c11);
  __syncthreads();
  if (threadIdx.x < 4 && s_cnt[threadIdx.x]) atomicAdd(&out[threadIdx.x], s_cnt[threadIdx.x]);
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

// held-out log-likelihood term of one (individual, location) entry (snp_likelihood,
// src/snpsamplinge.hh:336-360): log(max(C(2,y) q^y (1-q)^(2-y), 1e-30)), q = sum_k Ebeta_k Etheta_nk
__device__ __forceinline__ double heldout_term(const double *gam, uint32_t npad, uint32_t K, const double *lam_loc,
                                               uint32_t n, int x) {
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
  const double v = (x == 1) ? 2.0 : 1.0;  // 2!/(x!(2-x)!)
  double sum = v * pow(q, (double)x) * pow(1.0 - q, (double)(2 - x));
  if (sum < 1e-30) sum = 1e-30;
  return log(sum);
}

// The validation block of compute_likelihood (src/snpsamplinge.cc:476-498) for many locations in
// one launch: workgroup j handles requested location j -- its held-out entries are a span of the
// context's flat (ids, true genotypes) table; the terms are computed in parallel and then added
// by one thread in ascending individual order, the order of the reference's loop (and of
// the host-side sum this replaces), so the per-location sums are reproducible bit for bit.
struct HeldReq {
  unsigned long long start;  // first entry of the location in the flat table
  uint32_t len, loc;
};
__global__ __launch_bounds__(256) void ts_heldout_eval(const double *gam, uint32_t npad, uint32_t K, const double *lam,
                                                       const uint32_t *ids, const uint8_t *ytrue, const HeldReq *req,
                                                       double *terms, double *sums) {
  const HeldReq r = req[blockIdx.x];
  const double *lam_loc = lam + (size_t)r.loc * 2 * K;
  for (uint32_t e = threadIdx.x; e < r.len; e += 256u)
    terms[r.start + e] = heldout_term(gam, npad, K, lam_loc, ids[r.start + e], (int)ytrue[r.start + e]);
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    uint32_t e = 0;
    for (; e + 8u <= r.len; e += 8u) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = terms[r.start + e + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; e < r.len; ++e) s += terms[r.start + e];
    sums[blockIdx.x] = s;
  }
}

// ---------------------------------------------------------------------------
// On-box ceilings (tsamd_probe_stream): the memory traffic of the two pass kernels with
// nothing else in them, on the context's own arrays and launch geometry.
//  ts_probe_read: K row streams of 16-byte loads over a workgroup's chunk of pairs, forwards or
//                 backwards (the plain pass' weight reads).
//  ts_probe_rmw:  read w and gamma, write both back unchanged (the first pass' traffic).
__global__ __launch_bounds__(1024) void ts_probe_read(const double *w, uint32_t K, uint32_t npad, uint32_t chunk,
                                                      uint32_t rev, double *sink) {
  const uint32_t npairs = npad / 2;
  const uint32_t begin = blockIdx.x * chunk, end = min(begin + chunk, npairs);
  const uint32_t i0 = begin + threadIdx.x;
  const uint32_t cnt = (i0 < end) ? (end - i0 + blockDim.x - 1u) / blockDim.x : 0u;
  double acc = 0.0;
  for (uint32_t t = 0; t < cnt; ++t) {
    const uint32_t i = rev ? i0 + (cnt - 1u - t) * blockDim.x : i0 + t * blockDim.x;
#pragma unroll 8
    for (uint32_t k = 0; k < K; ++k) {
      const double2 v = reinterpret_cast<const double2 *>(w + (size_t)k * npad)[i];
      acc += v.x + v.y;
    }
  }
  if (acc == 123.456) sink[0] = acc;  // (keeps the loads alive)
}

__global__ __launch_bounds__(256) void ts_probe_rmw(double *w, double *gam, uint32_t K, uint32_t npad, uint32_t chunk,
                                                    double one, uint32_t think_ticks) {
  const uint32_t begin = blockIdx.x * chunk, end = min(begin + chunk, npad);
  for (uint32_t i = begin + threadIdx.x; i < end; i += 256u) {
    if (think_ticks) {  // stand-in for the arithmetic between an item's loads and stores (10 ns ticks): de-phases
      const unsigned long long t0 = wall_clock64();  // the workgroups' read and write bursts like the real kernel
      while (wall_clock64() - t0 < think_ticks) __builtin_amdgcn_s_sleep(2);
    }
    for (uint32_t k0 = 0; k0 < K; k0 += 8u) {
      double a[8], b[8];
#pragma unroll
      for (uint32_t q = 0; q < 8u; ++q) {
        const uint32_t k = min(k0 + q, K - 1u);
        a[q] = w[(size_t)k * npad + i];
        b[q] = gam[(size_t)k * npad + i];
      }
#pragma unroll
      for (uint32_t q = 0; q < 8u; ++q)
        if (k0 + q < K) {
          gam[(size_t)(k0 + q) * npad + i] = b[q] * one;  // one == 1.0: the state is unchanged, bit for bit
          w[(size_t)(k0 + q) * npad + i] = a[q] * one;
        }
    }
  }
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

// tsamd_debug_occupy: workgroups that hold a compute unit each for `ticks` (10 ns): 96 KB of LDS, so that no workgroup of
// the resident kernels (one per compute unit, ~140 KB of LDS) fits beside one.  Tells the host when it runs.
__global__ __launch_bounds__(256) void ts_occupy(unsigned long long ticks, unsigned long long *started) {
  __shared__ unsigned long long s_hold[12288];
  s_hold[threadIdx.x] = ticks;
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(started, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < s_hold[(threadIdx.x * 7u) % 256u]) __builtin_amdgcn_s_sleep(32);
}

}  // namespace tsamd
