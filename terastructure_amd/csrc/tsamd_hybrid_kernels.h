// ts_hybrid<K, WR, STREAM>: a WHOLE schedule in one launch for shards ABOVE the register capacity of ts_schedule (gfx950).
//
// ts_schedule (tsamd_resident_kernels.h) keeps the shard's N x K weights in the register file and gives up -- the
// context drops to ten launches per update -- as soon as one individual more is asked for (K = 20: 327 680 per GPU).
// The reference takes any -n / -k (src/main.cc:115-123).  This kernel keeps the same one-launch structure (gamma step,
// passes, in-launch exchange, epilogue; same State in, same State out) and splits a thread's individuals three ways:
//   * R = hy_reg_items(K) items in REGISTERS, as ts_schedule (one fewer above K = 20 and at K = 9);
//   * Q = hy_lds_items(K) items whose weights live in LDS for the whole launch (the 160 KB that ts_schedule spends on
//     half of gamma: a weight is read ten times per SNP, a gamma value once) -- K = 8: 9 items, K = 20: 3;
//   * every further item STREAMED: its weights are re-read from memory (Infinity Cache) every pass through a
//     software pipeline, its gamma step streams gamma, c_n and the weights and writes them back.
// gamma and c_n of ALL items stream from memory in the gamma step (one item ahead).  Capacity without streaming:
// 256 x 256 x (R + Q) individuals (K = 8: 1 638 400, K = 16: 786 432, K = 20: 524 288, K = 32: 327 680); above that
// the streamed part grows by 65 536 individuals per item.  Every SNP's last pass is exchanged on the spot (no deferral).
// Restated reference code: as ts_schedule -- PhiRunnerE::process / update_phimom / update_phidad / update_lambda_t
// (src/snpsamplinge.hh:276-300, :416-431, src/snpsamplinge.cc:742-759), update_gamma / estimate_theta (:695-740),
// update_lambda / estimate_beta (:267-296), optimize_lambda (:320-366).
#pragma once
#include "tsamd_resident_kernels.h"

namespace tsamd {

// items whose weights live in LDS: what 160 KB hold beside the K x 2 arrays, at most 16 (their codes share one register)
constexpr int hy_lds_items(int k) {
  const int n = (160 * 1024 - 1024 - 200 * k) / (k * 8 * 256);
  return n > 16 ? 16 : n;
}
// items in registers: ts_schedule's, one fewer above K = 20 (the streamed items' pipeline needs the registers); round 6, from the
// build's resource table (profiles/r06_kernel_resources.txt): K = 9 13 instead of 14 and K = 29 ... 32 one instead of two -- the
// streamed instantiations of those K used 20 ... 236 bytes of scratch
// (K = 22: floor(112 / 22) - 1 = 4, as before ts_schedule<22> went from 5 items to 4 in round 6)
constexpr int hy_reg_items(int k) { return k == 9 ? 13 : k <= 20 ? resident_items(k) : k <= 24 ? 112 / k - 1 : k <= 28 ? 2 : 1; }
// individuals a workgroup holds without streaming any weights
constexpr int hybrid_resident_capacity(int k) { return (hy_reg_items(k) + hy_lds_items(k)) * kResidentBlock; }
// streamed items per thread at most (a bound on the loop, not a register budget: 4M individuals per GPU at least)
constexpr int kHybridMaxStreamed = 64;

// update_gamma + update_rho_indiv (src/snpsamplinge.cc:688-719) for one individual with nodekappa = 0.5, then the new
// weights -- ts_schedule's lean form: (1 - rho) gamma + rho alpha + w_k (c0 sb_k0 + c1 sb_k1), rho * scale folded into
// c0 / c1, ONE reciprocal for both parents; an unobserved genotype takes the same instructions with a step of exactly 0.
template <int KT>
__device__ __forceinline__ void hy_gamma_one(double (&gx)[KT], double (&wx)[KT], uint32_t nib, uint32_t &cn, const double *s_sb, const DevParams &p) {
  // (nib: the genotype's factors (y, 2 - y) as code_nibble packs them; 0 when unobserved)
  const double mom = (double)(nib & 3u), dad = (double)((nib >> 2) & 3u);
  const bool ok = (nib & 15u) != 0u;
  double s0 = 0.0, s1 = 0.0;
  uint32_t zo = 0u;  // (opaque zero: exp(Elogbeta) is re-read from LDS where it is used, not held in 4K registers)
  asm volatile("" : "+v"(zo));
  const double *sbv = s_sb + zo;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    s0 = fma(wx[k], sbv[2 * k], s0);
    s1 = fma(wx[k], sbv[2 * k + 1], s1);
  }
  const double rho = ok ? fast_rsqrt(p.nodetau0 + (double)cn) : 0.0;
  const double inv = fast_rcp(s0 * s1) * (rho * p.gamma_scale);
  const double c0 = (mom * s1) * inv, c1 = (dad * s0) * inv, keep = 1.0 - rho, ra = rho * p.alpha;
#pragma unroll
  for (int k = 0; k < KT; ++k) gx[k] = fma(wx[k], fma(c0, sbv[2 * k], c1 * sbv[2 * k + 1]), fma(keep, gx[k], ra));
  if constexpr (KT <= 8) gamma_to_w<KT>(gx, wx); else gamma_to_w_lean<KT>(gx, wx);
  cn = ok ? cn + 1u : cn;
}

template <int KT, int WR, bool STREAM>
__global__ __launch_bounds__(256, 1) void ts_hybrid(Ctl *ctl_a, double *w_a, uint32_t npad_a, uint32_t chunk_a, uint32_t par_arg,
                                                    const uint32_t *sched, uint32_t n_sched, ResXchg *xb, uint32_t serial, const DevParams p) {
  constexpr int BLOCK = 256, kWaves = BLOCK / 64, R = hy_reg_items(KT), Q = hy_lds_items(KT), RQ = R + Q;
  // exp(Elogbeta) of a pass in registers / in scalar registers (vector registers only up to K = 12: above, the sweep also
  // holds two streamed items' rows in flight and reads the pairs from LDS at each use instead)
  constexpr bool BS = KT <= 12, BSC = KT <= 8;
  constexpr uint32_t J = 2 * KT;
  static_assert(resident_vec(KT) == 1 && R <= 16 && Q >= 1 && Q <= 16, "one individual per item; an item class's codes share one register");
  __shared__ __attribute__((aligned(16))) double s_eb[J];
  __shared__ __attribute__((aligned(16))) double s_sb[J];
  __shared__ double s_lam[J], s_diff[J], s_tot[res_tot_doubles<KT, WR>()], s_plam[J], s_peb[J];
  __shared__ double s_red[kWaves * J];
  __shared__ int s_alive[4];
  __shared__ double s_w[Q][KT][BLOCK];  // the weights of the LDS items
  const uint32_t par = par_arg & 1u;
  Ctl *ctl = ctl_a;
  const State *S = &ctl->st[par ^ 1u];
  State *W = &ctl->st[par];
  const uint32_t svalid = S->valid, sloc = S->loc, shol = S->hol, siters = S->iters, sidx = S->idx;
  const unsigned long long epoch_now = S->epoch + 1ull;
  const uint32_t xseq0 = ctl->xseq;
  uint32_t tid = threadIdx.x;
  const size_t np = npad_a;
  const uint32_t nitems = npad_a;
  const uint32_t begin = blockIdx.x * chunk_a, end = min(begin + chunk_a, nitems);
  uint32_t i0 = begin + tid;
  uint32_t cnt = (i0 < end) ? (end - i0 + BLOCK - 1u) / BLOCK : 0u;  // this thread's items, all classes
  // (see ts_schedule: the values every address derives from are made opaque per use, or a few hundred addresses are
  // hoisted out of the schedule loop and spilled)
  auto fresh = [&]() { asm volatile("" : "+v"(tid), "+v"(i0), "+v"(cnt)); };
  // item t of the thread, or its last one, or -- a thread that owns none -- the shard's last item
  auto item_or_last = [&](uint32_t t) { return min(i0, nitems - 1u) + min(t, max(cnt, 1u) - 1u) * BLOCK; };
  const uint32_t cnt_wg = begin < end ? (end - begin + BLOCK - 1u) / BLOCK : 0u;  // items any thread of the workgroup owns (uniform)
  // ... of which streamed (STREAM = false: the host has checked that there are none; the streamed code is compiled out)
  const uint32_t scnt_wg = (STREAM && cnt_wg > (uint32_t)RQ) ? cnt_wg - (uint32_t)RQ : 0u;
  const uint32_t g = blockIdx.x % (uint32_t)kResGroups, m = blockIdx.x / (uint32_t)kResGroups;

  if (__hip_atomic_load(&xb->abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) return;  // (see ts_resident)
  if (n_sched == 0u) {
    if (blockIdx.x == 0) carry_state(S, W, J);
    return;
  }
  if (tid < 4) s_alive[tid] = 1;
  __syncthreads();
  uint32_t xcount = 1u;
  // the entry exchange: empty rows, nothing modified yet.  All workgroups resident?
  if (!res_exchange<KT, WR>(xb, p, xseq0 + 1u, 1u, 0.0, g, m, gridDim.x, s_tot, s_alive, tid, fail_code(xseq0 + 1u, true, par, serial),
                            WR == 0 ? (unsigned long long)p.probe_ticks : kResWaitTicks))
    return;
  if constexpr (WR > 0) {
    // (sharded: the commit exchange of ts_schedule -- the entry exchange's verdict is not collective by itself)
    xcount += 1u;
    if (!res_exchange<KT, WR>(xb, p, xseq0 + xcount, 1u, 0.0, g, m, gridDim.x, s_tot, s_alive, tid, fail_code(xseq0 + xcount, true, par, serial),
                              kResWaitTicks / 3ull))
      return;
  }
  // the weights of the register and LDS items: loaded once, kept for the whole launch
  double buf[R][KT];
#pragma unroll
  for (int t = 0; t < R; ++t) {
#pragma unroll
    for (int k = 0; k < KT; ++k) buf[t][k] = (w_a + (size_t)k * np)[item_or_last((uint32_t)t)];
    if (t & 1) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const uint32_t i = item_or_last((uint32_t)(R + q));
#pragma unroll
    for (int k = 0; k < KT; ++k) s_w[q][k][tid] = (w_a + (size_t)k * np)[i];
  }
  // the genotypes of a column for this thread's register items (FIRST = 0, N = R) or LDS items (FIRST = R, N = Q), decoded
  // once per SNP into nibbles (code_nibble: (y, 2 - y), 0 when unobserved or not owned) and packed into two registers
  auto load_codes = [&](uint32_t loc_, auto first_c, auto n_c) -> uint2 {
    constexpr int FIRST = decltype(first_c)::value, N = decltype(n_c)::value;
    const uint32_t *col = reinterpret_cast<const uint32_t *>(p.bed + (size_t)loc_ * p.colstride);
    uint32_t word[N];
#pragma unroll
    for (int t = 0; t < N; ++t) word[t] = col[item_or_last((uint32_t)(FIRST + t)) / 16u];
    uint32_t out[2] = {0u, 0u};
#pragma unroll
    for (int t = 0; t < N; ++t) {
      const uint32_t i = item_or_last((uint32_t)(FIRST + t));
      const uint32_t c = (uint32_t)(FIRST + t) < cnt ? (word[t] >> (2u * (i % 16u))) & 3u : 1u;
      out[t / 8] |= code_nibble(c) << (4u * (uint32_t)(t % 8));
    }
    return make_uint2(out[0], out[1]);
  };
  using IC0 = std::integral_constant<int, 0>;
  using ICR = std::integral_constant<int, R>;
  using ICQ = std::integral_constant<int, Q>;
  uint2 pcodes = svalid ? load_codes(sloc, IC0{}, ICR{}) : make_uint2(0u, 0u), pcodes2 = svalid ? load_codes(sloc, ICR{}, ICQ{}) : make_uint2(0u, 0u);
  if (tid < J) {
    s_sb[tid] = S->eb[tid];
    s_plam[tid] = svalid ? p.lam[(size_t)sloc * J + tid] : 0.0;
    s_peb[tid] = svalid ? p.eb[(size_t)sloc * J + tid] : 0.0;
  }
  bool do_gamma = svalid != 0u && shol == 0u;
  bool prev_valid = svalid != 0u;
  uint32_t prev_loc = sloc, prev_hol = shol, prev_iters = siters;
  bool w_dirty = false;
  unsigned long long tp_run = ctl->total_passes;
  uint32_t last_it = ctl->last_iters;
  auto count_snp_deferred = [&](uint32_t its) {  // (as in ts_schedule)
    const uint32_t bin = min(its, (uint32_t)TSAMD_PASS_HIST_BINS - 1u);
    __hip_atomic_fetch_add(&ctl->pass_hist[bin], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tp_run += (unsigned long long)its;
    last_it = its;
  };
#ifdef TSAMD_SCHED_TIME  // diagnostic build (tools/variant.sh): where a SNP's time goes, 10 ns ticks, workgroup 0
  unsigned long long tk_gamma = 0, tk_sweep = 0, tk_xchg = 0, tk_epi = 0, tk_fold = 0, tk_mark = wall_clock64();
  const unsigned long long tk_start = tk_mark;
#define TSAMD_HK(acc)                               \
  do {                                              \
    const unsigned long long now_ = wall_clock64(); \
    acc += now_ - tk_mark;                          \
    tk_mark = now_;                                 \
  } while (0)
#else
#define TSAMD_HK(acc) \
  do {                \
  } while (0)
#endif
  __syncthreads();

  uint32_t iters = 0u;
  double lam_old = 0.0, eb_used = 0.0;
  double b0[BS ? KT : 1], b1[BS ? KT : 1], acc0[KT], acc1[KT];
  bool complete = false;
  uint2 codes = make_uint2(0u, 0u), codes2 = make_uint2(0u, 0u);
  uint32_t loc = 0u, hol = 0u;
  const uint32_t *col = nullptr;  // the running SNP's column (streamed items take their codes from it per pass)

  auto begin_pass = [&]() {
    fresh();
    iters += 1u;
    lam_old = s_lam[tid < J ? tid : 0u];
    eb_used = s_eb[tid < J ? tid : 0u];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      if constexpr (BSC) {
        b0[k] = uniform_f64(s_eb[2 * k]);
        b1[k] = uniform_f64(s_eb[2 * k + 1]);
      } else if constexpr (BS) {
        b0[k] = s_eb[2 * k];
        b1[k] = s_eb[2 * k + 1];
      }
      acc0[k] = acc1[k] = 0.0;
    }
  };
  auto consume = [&](const double (&wcur)[KT], uint32_t nib) {
    uint32_t zo = 0u;  // (opaque zero: the LDS reads of exp(Elogbeta) are repeated per item instead of held in 4K registers)
    if constexpr (!BS) asm volatile("" : "+v"(zo));
    res_consume_md<KT, BS>(wcur, (double)(nib & 3u), (double)(nib >> 2), b0, b1, reinterpret_cast<const double2 *>(s_eb) + zo, acc0, acc1);
  };
  // a streamed item's rows and its word of the column, from memory (clamped: static load counts)
  auto load_streamed = [&](uint32_t s, double (&wv)[KT], uint32_t &word) {
    const uint32_t i = item_or_last((uint32_t)RQ + s);
#pragma unroll
    for (int k = 0; k < KT; ++k) wv[k] = (w_a + (size_t)k * np)[i];
    word = col[i / 16u];
  };
  auto streamed_code = [&](uint32_t s, uint32_t word) -> uint32_t {
    const uint32_t i = item_or_last((uint32_t)RQ + s);
    return code_nibble((uint32_t)RQ + s < cnt ? (word >> (2u * (i % 16u))) & 3u : 1u);
  };
  // The streamed items of a pass go through two buffers, one item ahead.  The FIRST item of a pass is requested ahead of the
  // pass: for a SNP's first pass right after the gamma step (which changes the weights), for every later one while the
  // workgroup waits in the exchange (res_exchange's overlap hook: the weights do not change between the passes of a SNP,
  // and the memory system idles there otherwise).  One buffer only is in flight across the exchange: with both, the
  // register items' sweep no longer fits the register file.
  double sa[KT], sbuf[KT];
  uint32_t worda = 0u, wordb = 0u;
  auto request_first = [&]() { load_streamed(0u, sa, worda); };
  auto sweep = [&]() {
#pragma unroll
    for (int t = 0; t < R; ++t) {
      if ((uint32_t)t >= cnt_wg) continue;
      fresh();
      double wcur[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) wcur[k] = buf[t][k];
      consume(wcur, nibble_of(codes, t));
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if ((uint32_t)(R + q) >= cnt_wg) continue;
      fresh();
      double wcur[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) wcur[k] = s_w[q][k][tid];
      consume(wcur, nibble_of(codes2, q));
      __builtin_amdgcn_sched_barrier(0);
    }
    // streamed items: two-stage software pipeline, two items per turn (no register moves between the stages; clamped
    // requests: static load counts)
    for (uint32_t s = 0; STREAM && s < scnt_wg; s += 2u) {
      fresh();
      load_streamed(s + 1u, sbuf, wordb);
      __builtin_amdgcn_sched_barrier(0);
      consume(sa, streamed_code(s, worda));
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1u >= scnt_wg) break;
      fresh();
      load_streamed(s + 2u, sa, worda);
      __builtin_amdgcn_sched_barrier(0);
      consume(sbuf, streamed_code(s + 1u, wordb));
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // end of a pass: workgroup reduction, the in-launch exchange, the K x 2 epilogue (threads < 2K, shared arrays).
  // false: the exchange gave up.
  auto finish_pass = [&]() -> bool {
    fresh();
    res_fold<KT>(acc0, acc1, s_red, tid);
    __syncthreads();
    TSAMD_HK(tk_fold);
    double mine = 0.0;
    if (tid < J) {
      mine = s_red[tid];
#pragma unroll
      for (int wv = 1; wv < kWaves; ++wv) mine += s_red[wv * J + tid];
    }
    xcount += 1u;
    const uint32_t tag = xseq0 + xcount;
    // (under the pass cap this may be the SNP's last pass: then the next sweep follows a gamma step and requests afresh)
    const bool ahead = STREAM && iters < p.max_inner;
    auto overlap = [&]() {
      if constexpr (STREAM)
        if (ahead) request_first();
    };
    if (!res_exchange<KT, WR, kResOneLevelGrid, ResLay<KT>>(xb, p, tag, 1u, mine, g, m, gridDim.x, s_tot, s_alive, tid, fail_code(tag, false, par, serial),
                                                            kResWaitTicks, overlap))
      return false;
    TSAMD_HK(tk_xchg);
    if (tid < J) epilogue_values(p, res_total<KT, WR>(s_tot, tid), eb_used, lam_old, s_lam, s_eb, s_diff);
    __syncthreads();
    complete = epilogue_complete(p, iters, J, s_diff);
    TSAMD_HK(tk_epi);
    return true;
  };

  for (uint32_t idx = 0; idx < n_sched; ++idx) {
    const uint32_t ent = sched[idx];
    loc = ent & 0x7fffffffu;
    hol = ent >> 31;
    fresh();
    // lambda / exp(Elogbeta) of the location: a SNP at the location of its predecessor takes that one's final values from
    // LDS; anything older was published by workgroup 0 (agent-scope stores, out before it joined the exchange that
    // everybody has completed since)
    if (tid < J) {
      const bool local = prev_valid && loc == prev_loc;
      const double gl = __hip_atomic_load(&p.lam[(size_t)loc * J + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double ge = __hip_atomic_load(&p.eb[(size_t)loc * J + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_lam[tid] = local ? s_plam[tid] : gl;
      s_eb[tid] = local ? s_peb[tid] : ge;
    }
    codes = load_codes(loc, IC0{}, ICR{});
    codes2 = load_codes(loc, ICR{}, ICQ{});
    iters = 0u;
    // ---- the previous SNP's gamma step: phi from the weights as they are and the exp(Elogbeta) of that SNP's last
    // pass (s_sb); gamma, c_n -- and a streamed item's weights -- come from memory one item ahead and go back at once
    if (do_gamma) {
      const uint32_t *pcol = reinterpret_cast<const uint32_t *>(p.bed + (size_t)prev_loc * p.colstride);
      double gs[KT];
      uint32_t cs = 0u;
      auto load_gamma = [&](uint32_t i, double (&gq)[KT], uint32_t &cq) {
#pragma unroll
        for (int k = 0; k < KT; ++k) gq[k] = (p.gam + (size_t)k * np)[i];
        cq = p.cnt[i];
      };
      auto store_gamma = [&](uint32_t i, const double (&gq)[KT], uint32_t cq) {
#pragma unroll
        for (int k = 0; k < KT; ++k) (p.gam + (size_t)k * np)[i] = gq[k];
        p.cnt[i] = cq;
      };
      load_gamma(item_or_last(0u), gs, cs);
#pragma unroll
      for (int t = 0; t < RQ; ++t) {
        if ((uint32_t)t >= cnt_wg) continue;
        fresh();
        const uint32_t i = item_or_last((uint32_t)t);
        double gv[KT];
        uint32_t cv = cs;
#pragma unroll
        for (int k = 0; k < KT; ++k) gv[k] = gs[k];
        if ((uint32_t)(t + 1) < cnt_wg && t + 1 < RQ) load_gamma(item_or_last((uint32_t)t + 1u), gs, cs);
        __builtin_amdgcn_sched_barrier(0);
        double wcur[KT];
        if (t < R) {
#pragma unroll
          for (int k = 0; k < KT; ++k) wcur[k] = buf[t < R ? t : 0][k];
        } else {
#pragma unroll
          for (int k = 0; k < KT; ++k) wcur[k] = s_w[t < R ? 0 : t - R][k][tid];
        }
        const uint32_t pcode = t < R ? nibble_of(pcodes, t < R ? t : 0) : nibble_of(pcodes2, t < R ? 0 : t - R);
        hy_gamma_one<KT>(gv, wcur, pcode, cv, s_sb, p);
        if ((uint32_t)t < cnt) store_gamma(i, gv, cv);
        if (t < R) {
#pragma unroll
          for (int k = 0; k < KT; ++k) buf[t < R ? t : 0][k] = wcur[k];
        } else {
#pragma unroll
          for (int k = 0; k < KT; ++k) s_w[t < R ? 0 : t - R][k][tid] = wcur[k];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (STREAM && scnt_wg > 0u) {
        double wsn[KT];
        uint32_t pwn = 0u;
        auto load_sitem = [&](uint32_t s) {
          const uint32_t i = item_or_last((uint32_t)RQ + s);
          load_gamma(i, gs, cs);
#pragma unroll
          for (int k = 0; k < KT; ++k) wsn[k] = (w_a + (size_t)k * np)[i];
          pwn = pcol[i / 16u];
        };
        load_sitem(0u);
        for (uint32_t s = 0; s < scnt_wg; ++s) {
          fresh();
          const uint32_t i = item_or_last((uint32_t)RQ + s);
          double gv[KT], wcur[KT];
          uint32_t cv = cs;
          const uint32_t pw = pwn;
#pragma unroll
          for (int k = 0; k < KT; ++k) {
            gv[k] = gs[k];
            wcur[k] = wsn[k];
          }
          load_sitem(min(s + 1u, scnt_wg - 1u));
          __builtin_amdgcn_sched_barrier(0);
          const uint32_t pcode = code_nibble((uint32_t)RQ + s < cnt ? (pw >> (2u * (i % 16u))) & 3u : 1u);
          hy_gamma_one<KT>(gv, wcur, pcode, cv, s_sb, p);
          if ((uint32_t)RQ + s < cnt) {
            store_gamma(i, gv, cv);
#pragma unroll
            for (int k = 0; k < KT; ++k) (w_a + (size_t)k * np)[i] = wcur[k];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // (the sweeps of this SNP re-read the streamed weights: this thread's own stores, same addresses -- in order)
      }
      w_dirty = true;
    }
    col = reinterpret_cast<const uint32_t *>(p.bed + (size_t)loc * p.colstride);
    if constexpr (STREAM) request_first();  // (the first pass' first streamed item: its latency hides behind the other items' sweep)
    __syncthreads();
    TSAMD_HK(tk_gamma);
    // ---- the passes of the new SNP ------------------------------------------------------------------------------
    complete = false;
    while (!complete) {  // (a pass that follows another: the pass cap was not reached, so the exchange's overlap hook has requested its first streamed item)
      begin_pass();
      sweep();
      TSAMD_HK(tk_sweep);
      if (!finish_pass()) return;
    }
    // ---- the SNP is complete: s_lam / s_eb hold its final values, eb_used the exp(Elogbeta) its last pass used.
    // Workgroup 0 publishes (out before it joins the next exchange); everybody keeps what the next gamma step needs.
    const double fin_lam = s_lam[tid < J ? tid : 0u], fin_eb = s_eb[tid < J ? tid : 0u];
    if (blockIdx.x == 0) {
      if (tid < J) {
        __hip_atomic_store(&p.lam[(size_t)loc * J + tid], fin_lam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&p.eb[(size_t)loc * J + tid], fin_eb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (tid == 0) count_snp_deferred(iters);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid < J) {
      s_sb[tid] = eb_used;
      s_plam[tid] = fin_lam;
      s_peb[tid] = fin_eb;
    }
    pcodes = codes;
    pcodes2 = codes2;
    do_gamma = hol == 0u;
    prev_valid = true;
    prev_loc = loc;
    prev_hol = hol;
    prev_iters = iters;
    __syncthreads();
  }

  // ---- end of the launch: the register and LDS items' weights go back to memory, the state to the next call --------
  if (w_dirty) {
#pragma unroll
    for (int t = 0; t < RQ; ++t) {
      fresh();
      if ((uint32_t)t < cnt) {
        const uint32_t i = i0 + (uint32_t)t * BLOCK;
#pragma unroll
        for (int k = 0; k < KT; ++k) (w_a + (size_t)k * np)[i] = t < R ? buf[t < R ? t : 0][k] : s_w[t < R ? 0 : t - R][k][tid];
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
      ctl->total_passes = tp_run;
      ctl->last_iters = last_it;
      if (p.host_error) {
        __hip_atomic_store(p.host_error + 1, (unsigned long long)last_it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(p.host_error + 2, tp_run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the histogram atomics of this thread have landed)
#ifdef TSAMD_SCHED_TIME
      if (n_sched >= 16u)
        printf("ts_hybrid n=%u exchanges=%u streamed items %u | per SNP (us): gamma %.2f sweeps %.2f folds %.2f exchanges %.2f epilogues %.2f | "
               "whole launch %.1f us\n", n_sched, xcount, scnt_wg, tk_gamma * 0.01 / n_sched, tk_sweep * 0.01 / n_sched, tk_fold * 0.01 / n_sched,
               tk_xchg * 0.01 / n_sched, tk_epi * 0.01 / n_sched, (wall_clock64() - tk_start) * 0.01);
#endif
    }
    __syncthreads();
    if (p.host_error && tid < (uint32_t)TSAMD_PASS_HIST_BINS)
      __hip_atomic_store(p.host_error + 3 + tid, __hip_atomic_load(&ctl->pass_hist[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
#undef TSAMD_HK
}

}  // namespace tsamd
