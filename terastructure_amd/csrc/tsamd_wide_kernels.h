// Wide-K fallback (TSAMD_SPECIALIZED_K < K <= TSAMD_MAX_K): the same state machine and
// arithmetic as ts_pass (tsamd_kernels.h) with K as a run-time value.  A thread owns single
// individuals; the 2K accumulators do not fit in registers, so the weights are read twice
// per pass: once for the two normalisers S0/S1 of each of the thread's items, then again in
// chunks of kWideChunk populations to accumulate, fold and store that chunk.  Slower than
// the K-specialised kernels by design; it exists so that every K the reference accepts runs.
// Included by tsamd.hip only.
#pragma once
#include "tsamd_kernels.h"

namespace tsamd {

constexpr int kWideBlock = 512;
constexpr int kWideItems = 8;   // max individuals per thread (the launch geometry guarantees it)
constexpr int kWideChunk = 8;   // populations accumulated per register chunk
constexpr int kWideJ = 2 * TSAMD_MAX_K;

template <bool FIRST>
__global__ __launch_bounds__(kWideBlock) void ts_pass_wide(DevParams p, uint32_t par_arg, uint32_t nrows_hint) {
  const uint32_t par = par_arg & 1u;  // (bit 1, the sweep direction of ts_pass, does not apply here)
  constexpr int BLOCK = kWideBlock;
  constexpr int kWaves = BLOCK / 64;
  __shared__ double s_eb[kWideJ];
  __shared__ double s_lam[kWideJ];
  __shared__ double s_sb[kWideJ];
  __shared__ double s_plam[kWideJ];
  __shared__ double s_peb[kWideJ];
  __shared__ double s_diff[kWideJ];
  __shared__ double s_red[kWaves][kWideJ];
  __shared__ double s_fin[BLOCK];

  Ctl *ctl = p.ctl;
  const State *S = &ctl->st[par ^ 1u];
  State *W = &ctl->st[par];
  const uint32_t tid = threadIdx.x;
  const uint32_t K = p.K, J = 2 * K;
  const size_t np = p.npad;
  const uint32_t chunk = p.chunk_first;  // individuals per workgroup (both kernels of the wide path)
  const uint32_t begin = blockIdx.x * chunk;
  const uint32_t end = min(begin + chunk, p.npad);
  (void)nrows_hint;

  const uint32_t sidx = S->idx, svalid = S->valid, sdone = S->done, sloc = S->loc, shol = S->hol;
  const uint32_t siters = S->iters, snrows = S->nrows;
  const uint32_t sched_len = ctl->sched_len;
  const uint32_t *sched = ctl->sched;
  const PendingIn pin = load_pending(S, J);
  const double *rowsR = p.xchg_world  ? p.xchg->rows[par ^ 1u]
                        : p.rows_from_lt ? ctl->lt_sum[par ^ 1u]
                                         : p.partials + (size_t)(par ^ 1u) * kMaxGrid * J;
  const uint32_t nrowsR = p.xchg_world ? p.xchg_world * snrows : p.rows_from_lt ? 1u : snrows;
  double *rowsW = p.partials + (size_t)par * kMaxGrid * J;

  const bool pending = svalid != 0u && sdone == 0u;
  const unsigned long long epoch_now = S->epoch + 1ull;
  if (p.xchg_world) {
    publish_progress(p, epoch_now);
    if (pending) {
      wait_peer_rows(p, par ^ 1u, S->epoch, nrowsR);
      xchg_test_stall(p);
    }
  }
  double vrow = 0.0;
  if (pending) vrow = row_partial_sum<BLOCK>(rowsR, nrowsR, J);
  uint32_t loc, hol, idx, iters;
  bool do_gamma = false;
  uint32_t prev_loc = 0;

  if constexpr (!FIRST) {
    if (!pending) {
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
    idx = sidx + 1u;
    if (pending) finish_pending<BLOCK>(p, pin, vrow, J, s_fin, s_plam, s_peb, s_diff);
    if (idx >= sched_len) {
      if (blockIdx.x == 0) {
        if (pending)
          publish_complete(p, ctl, S, W, J, s_plam, s_peb, true);
        else
          carry_state(S, W, J);
      }
      return;
    }
    const uint32_t ent = sched[idx];
    loc = ent & 0x7fffffffu;
    hol = ent >> 31;
    iters = 1u;
    do_gamma = svalid != 0u && shol == 0u;
    prev_loc = sloc;
    if (tid < J) {
      s_sb[tid] = S->eb[tid];
      if (pending && sloc == loc) {
        s_lam[tid] = s_plam[tid];
        s_eb[tid] = s_peb[tid];
      } else {
        s_lam[tid] = p.lam[(size_t)loc * J + tid];
        s_eb[tid] = p.eb[(size_t)loc * J + tid];
      }
    }
    __syncthreads();
  }

  const uint32_t *col = reinterpret_cast<const uint32_t *>(p.bed + (size_t)loc * p.colstride);
  const uint32_t *pcol = reinterpret_cast<const uint32_t *>(p.bed + (size_t)prev_loc * p.colstride);
  const uint32_t i0 = begin + tid;
  const uint32_t cnt = (i0 < end) ? min((end - i0 + BLOCK - 1u) / BLOCK, (uint32_t)kWideItems) : 0u;

  // ---- FIRST: the previous SNP's gamma step for this thread's individuals ------------------
  if (FIRST && do_gamma) {
    for (uint32_t t = 0; t < cnt; ++t) {
      const uint32_t i = i0 + t * BLOCK;
      const uint32_t pcode = (pcol[i >> 4] >> (2u * (i & 15u))) & 3u;
      double mom, dad;
      bool ok;
      code_weights(pcode, mom, dad, ok);
      if (!ok) continue;
      double s0 = 0.0, s1 = 0.0;
      for (uint32_t k = 0; k < K; ++k) {
        const double wk = p.w[(size_t)k * np + i];
        s0 = fma(wk, s_sb[2 * k], s0);
        s1 = fma(wk, s_sb[2 * k + 1], s1);
      }
      uint32_t c = p.cnt[i];
      const double base = p.nodetau0 + (double)c;
      const double rho = (p.nodekappa == 0.5) ? 1.0 / sqrt(base) : pow(base, -p.nodekappa);
      p.cnt[i] = c + 1u;
      const double c0 = mom / s0, c1 = dad / s1;
      double amax = -1.0e300;
      for (uint32_t k = 0; k < K; ++k) {
        const double wk = p.w[(size_t)k * np + i];
        const double e = c0 * (wk * s_sb[2 * k]) + c1 * (wk * s_sb[2 * k + 1]);
        double g = p.gam[(size_t)k * np + i];
        g += rho * (p.alpha + p.gamma_scale * e - g);
        p.gam[(size_t)k * np + i] = g;
        double z, a;
        exp_digamma_split(g, z, a);
        amax = fmax(amax, a);
      }
      for (uint32_t k = 0; k < K; ++k) {  // (own stores above: same thread, program order)
        double z, a;
        exp_digamma_split(p.gam[(size_t)k * np + i], z, a);
        p.w[(size_t)k * np + i] = z * exp_nonpos(a - amax);
      }
    }
  }

  // ---- pass, phase A: the two normalisers of each of the thread's individuals ---------------
  double c0[kWideItems], c1[kWideItems];
#pragma unroll
  for (int t = 0; t < kWideItems; ++t) {
    c0[t] = c1[t] = 0.0;
    if ((uint32_t)t < cnt) {
      const uint32_t i = i0 + (uint32_t)t * BLOCK;
      const uint32_t code = (col[i >> 4] >> (2u * (i & 15u))) & 3u;
      double mom, dad;
      bool ok;
      code_weights(code, mom, dad, ok);
      double s0 = 0.0, s1 = 0.0;
      for (uint32_t k = 0; k < K; ++k) {
        const double wk = p.w[(size_t)k * np + i];
        s0 = fma(wk, s_eb[2 * k], s0);
        s1 = fma(wk, s_eb[2 * k + 1], s1);
      }
      c0[t] = mom / s0;
      c1[t] = dad / s1;
    }
  }

  // ---- phase B: accumulate kWideChunk populations at a time, fold, next chunk ---------------
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  for (uint32_t k0 = 0; k0 < K; k0 += kWideChunk) {
    double a0[kWideChunk], a1[kWideChunk];
#pragma unroll
    for (int q = 0; q < kWideChunk; ++q) a0[q] = a1[q] = 0.0;
#pragma unroll
    for (int t = 0; t < kWideItems; ++t)
      if ((uint32_t)t < cnt) {
        const uint32_t i = i0 + (uint32_t)t * BLOCK;
#pragma unroll
        for (int q = 0; q < kWideChunk; ++q)
          if (k0 + q < K) {
            const double wk = p.w[(size_t)(k0 + q) * np + i];
            a0[q] = fma(c0[t], wk, a0[q]);
            a1[q] = fma(c1[t], wk, a1[q]);
          }
      }
    using Fold = WaveFold<2 * kWideChunk>;
    double v[Fold::P];
#pragma unroll
    for (int q = 0; q < kWideChunk; ++q) {
      v[2 * q] = a0[q];
      v[2 * q + 1] = a1[q];
    }
    const double tot = Fold::fold(v, lane);
    const int slot = Fold::slot(lane);
    constexpr uint32_t kRep = 64 / Fold::P;
    if ((lane & (kRep - 1u)) == 0u && 2 * k0 + (uint32_t)slot < J) s_red[wave][2 * k0 + slot] = tot;
  }
  __syncthreads();
  {
    double v = 0.0;
    if (tid < J) {
      v = s_red[0][tid];
      for (int wv = 1; wv < kWaves; ++wv) v += s_red[wv][tid];
    }
    store_row(p, par, epoch_now, v, J, rowsW, FIRST && !pending);
  }

  if (blockIdx.x == 0) {
    if (FIRST && pending) publish_complete(p, ctl, S, W, J, s_plam, s_peb, false);
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
}

// w = exp(psi(gamma)) * const for every individual (tsamd_set_gamma), run-time K
__global__ __launch_bounds__(256) void ts_refresh_w_wide(DevParams p) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= p.npad) return;
  const size_t np = p.npad;
  double amax = -1.0e300;
  for (uint32_t k = 0; k < p.K; ++k) {
    double z, a;
    exp_digamma_split(p.gam[(size_t)k * np + i], z, a);
    amax = fmax(amax, a);
  }
  for (uint32_t k = 0; k < p.K; ++k) {
    double z, a;
    exp_digamma_split(p.gam[(size_t)k * np + i], z, a);
    p.w[(size_t)k * np + i] = z * exp_nonpos(a - amax);
  }
}

}  // namespace tsamd
