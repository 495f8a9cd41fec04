// ts_hybhol<K>: the batched validation block for shards ABOVE the register capacity of ts_schedule (gfx950).
//
// compute_likelihood (src/snpsamplinge.cc:476-498) runs optimize_lambda once per validation location with _hol_mode set;
// the workers then skip the gamma step (src/snpsamplinge.cc:660-668): theta is frozen for the whole block and distinct
// locations are independent.  ts_holblock (tsamd_holblock_kernels.h) runs such a block BX locations at a time for shards whose
// weights fit the register file.  A context that runs ts_hybrid -- N = 1M, K = 20 on one GPU: 160 MB of weights against a
// 128 MB register file -- ran the block entry by entry until round 5: 191 us per location, ten sweeps each re-reading the
// streamed half of the weights, ten exchanges.  This kernel batches it (118 us):
//   * nothing is modified in validation mode and no gamma is needed, so the residency split is chosen for THIS kernel:
//     hh_reg_items(K) items of a thread in registers, hh_lds_items(K) in LDS (what the 160 KB hold beside the batch's
//     arrays), every further item streamed from memory (Infinity Cache) through two buffers, one item ahead;
//   * a pass sweeps the weights once per SUB-BATCH of BA = hh_sub(K) locations (K = 20: two), so a streamed item is
//     read once for BA locations -- the memory traffic per location-pass drops by that factor -- and the first streamed
//     item of a sub-batch is requested before its on-chip items are swept;
//   * ONE in-launch exchange per pass carries the rows of all BX = hh_batch(K) locations (WideLay), the BX epilogues run
//     side by side, a finished location is published and leaves the batch -- as in ts_holblock.
// Every per-location sum keeps the one-location path's order (ts_hybrid: a thread's items in index order whatever holds
// their weights, the same wave fold, the waves in order, the same member / group order in the exchange), so lambda,
// exp(Elogbeta), pass counts and the State left behind equal ts_hybrid's for the same entries BIT FOR BIT
// (tests/test_gpu_hybhol.py).  Launch geometry: ts_hybrid's (all workgroups, equal shares).  WR > 0: one launch per rank of a
// sharded run that runs ts_hybrid (up to 4 ranks), level 2 of the exchanges across the ranks (wide rows in Xchg::res_wide).
// Restated reference code: as ts_holblock.
#pragma once
#include "tsamd_holblock_kernels.h"
#include "tsamd_hybrid_kernels.h"

namespace tsamd {

// locations per sweep: as many as the register file takes beside a useful number of register items
// (three at K = 20 would cut the streamed bytes per location by a third, but the third location's accumulators live in AGPRs:
// 600 instructions per item, 185 of them register moves -- 174 against 118 us per location, with or without register items,
// profiles/r05_experiments.md)
#ifdef TSAMD_HH_SUB  // (experiments)
constexpr int hh_sub(int) { return TSAMD_HH_SUB; }
#else
constexpr int hh_sub(int k) { return k <= 4 ? 4 : k <= 20 ? 2 : 1; }
#endif
// exp(Elogbeta) of the sub-batch in vector registers for the sweep (else read as pairs from LDS at each use)
constexpr bool hh_bs(int k) { return k <= 8; }
// locations per exchange: a multiple of that, at most 16, rows of at most 256 values
constexpr int hh_batch(int k) {
  const int ba = hh_sub(k);
  int n = 128 / (k * ba);
  if (n > 16 / ba) n = 16 / ba;
  if (n < 1) n = 1;
  return n * ba;
}
// items in registers: what the budget (in doubles; tuned per range of K with tools/kcompile.sh UNIT=hhol, like
// resident_items) holds beside the accumulators, the pairs (K <= 8) and two streamed items' rows
#ifdef TSAMD_HH_BUDGET  // (experiments)
constexpr int hh_budget(int) { return TSAMD_HH_BUDGET; }
#else
constexpr int hh_budget(int k) {
  return k <= 8 ? 190 : k <= 10 ? 140 : k <= 13 ? 150 : k <= 14 ? 160 : k <= 15 ? 175 : k <= 16 ? 190 : k <= 20 ? 185 : k <= 23 ? 140 : k <= 24 ? 165 : k <= 27 ? 130 : k <= 28 ? 165 : 100;
}
#endif
constexpr int hh_reg_items(int k) {
  const int ba = hh_sub(k);
  const int fixed = ba * 2 * k + (hh_bs(k) ? ba * 2 * k : 0) + 2 * k;
  int r = (hh_budget(k) - fixed) / k;
  return r > 16 ? 16 : r < 0 ? 0 : r;  // (K >= 29: none -- the accumulators and the stream buffers take the budget; the on-chip items are the LDS ones)
}
// bytes of LDS the batch's arrays take (codes, rows, pairs, differences, totals, small arrays)
constexpr int hh_batch_lds(int k) {
  const int bx = hh_batch(k), j = 2 * k, jx = bx * j;
  return bx * 256 * 8 + bx * 4 * j * 8 + 2 * bx * j * 8 + (jx > 4 * j ? jx : 4 * j) * 8 + bx * 4 + 1024;
}
// items whose weights live in LDS: the rest of the 160 KB; the on-chip items' codes share one 16-nibble pair of registers
constexpr int hh_lds_items(int k) {
  const int n = (160 * 1024 - hh_batch_lds(k)) / (k * 8 * 256), room = 16 - hh_reg_items(k);
  return n < room ? n : room;
}

template <int KT, int WR>
__global__ __launch_bounds__(256, 1) void ts_hybhol(Ctl *ctl_a, const double *w_a, uint32_t npad_a, uint32_t chunk_a, uint32_t par_arg,
                                                    const uint32_t *sched, uint32_t n_sched, ResXchg *xb, uint32_t serial, const DevParams p) {
  constexpr int BLOCK = 256, kWaves = BLOCK / 64, R = hh_reg_items(KT), Q = hh_lds_items(KT), RQ = R + Q;
  constexpr int BA = hh_sub(KT), BX = hh_batch(KT), NSUB = BX / BA, KX = BX * KT;
  constexpr uint32_t J = 2 * KT, JX = 2 * KX;
  constexpr bool BS = hh_bs(KT);
  static_assert(resident_vec(KT) == 1 && RQ <= 16 && RQ >= 1, "one individual per item; the on-chip items' codes share two registers");
  static_assert(JX <= (uint32_t)BLOCK && BX <= 16 && BX % BA == 0, "a batch's row is brought by one thread per value");
  using Wide = WideLay<KX, KT>;
  __shared__ __attribute__((aligned(16))) double s_eb[BX][J];  // exp(Elogbeta) the running pass uses, per location of the batch
  __shared__ double s_diff[BX][J];
  __shared__ double s_tot[JX > 4 * J ? JX : 4 * J];
  __shared__ double s_red[BX][kWaves][J];
  __shared__ uint2 s_codes[BX][BLOCK];  // the ON-CHIP items' genotype factors of the batch's columns (code_nibble, item t in bits 4t .. 4t+3)
  __shared__ uint32_t s_loc[BX];        // the batch's locations (streamed items take their codes from the columns, per pass)
  __shared__ int s_alive[4];
  __shared__ double s_w[Q > 0 ? Q : 1][KT][BLOCK];  // the weights of the LDS items

  const uint32_t par = par_arg & 1u;
  Ctl *ctl = ctl_a;
  const State *S = &ctl->st[par ^ 1u];
  State *W = &ctl->st[par];
  const uint32_t sidx = S->idx;
  const unsigned long long epoch_now = S->epoch + 1ull;
  const uint32_t xseq0 = ctl->xseq;
  uint32_t tid = threadIdx.x;
  const size_t np = npad_a;
  const uint32_t nitems = npad_a;
  const uint32_t begin = blockIdx.x * chunk_a, end = min(begin + chunk_a, nitems);
  uint32_t i0 = begin + tid;
  uint32_t cnt = (i0 < end) ? (end - i0 + BLOCK - 1u) / BLOCK : 0u;  // this thread's items, all classes (ts_hybrid's geometry)
  auto fresh = [&]() { asm volatile("" : "+v"(tid), "+v"(i0), "+v"(cnt)); };
  auto item_or_last = [&](uint32_t t) { return min(i0, nitems - 1u) + min(t, max(cnt, 1u) - 1u) * BLOCK; };
  const uint32_t cnt_wg = begin < end ? (end - begin + BLOCK - 1u) / BLOCK : 0u;  // items any thread of the workgroup owns (uniform)
  const uint32_t scnt_wg = cnt_wg > (uint32_t)RQ ? cnt_wg - (uint32_t)RQ : 0u;    // ... of which streamed
  const uint32_t g = blockIdx.x % (uint32_t)kResGroups, m = blockIdx.x / (uint32_t)kResGroups;

  if (__hip_atomic_load(&xb->abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) return;  // (see ts_resident)
  if (n_sched == 0u) {
    if (blockIdx.x == 0) carry_state(S, W, J);
    return;
  }
  if (S->valid != 0u && S->hol == 0u) {
    // a training update's gamma step is pending: the host routes the first validation entry through ts_hybrid, which applies
    // it, before it launches this kernel -- anything else is a bug there; refuse loudly instead of dropping the step
    if (blockIdx.x == 0 && tid == 0) {
      const unsigned long long code = fail_code(0xffffffffu, false, par, serial);
      __hip_atomic_store(&xb->abort_word, 0xffffffffull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (p.host_error) {
        __hip_atomic_store(p.host_error + kHostDirtyWord, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(p.host_error, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }
  if (tid < 4) s_alive[tid] = 1;
  __syncthreads();
  uint32_t xcount = 1u;
  // the entry exchange: empty rows, nothing modified yet.  All workgroups resident?  (ts_hybrid's, in its layout)
  if (!res_exchange<KT, WR>(xb, p, xseq0 + 1u, 1u, 0.0, g, m, gridDim.x, s_tot, s_alive, tid, fail_code(xseq0 + 1u, true, par, serial),
                            WR == 0 ? (unsigned long long)p.probe_ticks : kResWaitTicks))
    return;
  if constexpr (WR > 0) {  // (sharded: the entry's verdict is committed by a second empty exchange, as in ts_schedule)
    xcount += 1u;
    if (!res_exchange<KT, WR>(xb, p, xseq0 + xcount, 1u, 0.0, g, m, gridDim.x, s_tot, s_alive, tid, fail_code(xseq0 + xcount, true, par, serial),
                              kResWaitTicks / 3ull))
      return;
  }
  // the weights of the register and LDS items: loaded once, never modified (theta is frozen in validation mode)
  double buf[R > 0 ? R : 1][KT];
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
  unsigned long long tp_run = ctl->total_passes;
#ifdef TSAMD_SCHED_TIME  // diagnostic build (tools/variant.sh UNIT=hhol): where a batch's time goes, 10 ns ticks, workgroup 0
  unsigned long long tk_setup = 0, tk_sweep = 0, tk_xchg = 0, tk_epi = 0, tk_mark = wall_clock64();
  uint32_t tk_passes = 0u;
  const unsigned long long tk_start = tk_mark;
#define TSAMD_BK(acc)                               \
  do {                                              \
    const unsigned long long now_ = wall_clock64(); \
    acc += now_ - tk_mark;                          \
    tk_mark = now_;                                 \
  } while (0)
#else
#define TSAMD_BK(acc) \
  do {                \
  } while (0)
#endif
  auto count_snp_deferred = [&](uint32_t its) {  // (as in ts_schedule: fire-and-forget histogram bump, totals in registers)
    const uint32_t bin = min(its, (uint32_t)TSAMD_PASS_HIST_BINS - 1u);
    __hip_atomic_fetch_add(&ctl->pass_hist[bin], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tp_run += (unsigned long long)its;
  };

  for (uint32_t base = 0; base < n_sched; base += (uint32_t)BX) {
    const uint32_t nb = min((uint32_t)BX, n_sched - base);
    fresh();
    // this thread's (location of the batch, value of its row)
    uint32_t vb = tid / J, vj = tid % J;
    const bool vmine = tid < JX && vb < nb;
    const uint32_t vloc = vmine ? sched[base + vb] & 0x7fffffffu : 0u;
    double lam_old = 0.0, eb_used = 0.0, eb_ran = 0.0;  // lambda before the pending epilogue, exp(Elogbeta) of the next / the last executed pass
    uint32_t it_mine = 0u;
    if (vmine) {
      lam_old = p.lam[(size_t)vloc * J + vj];
      eb_used = p.eb[(size_t)vloc * J + vj];
      s_eb[vb][vj] = eb_used;
    } else if (tid < JX) {
      s_eb[vb][vj] = 1.0;  // (a partial batch: the missing locations' rows are swept with their sub-batch, on finite values, and never read)
    }
    if (tid < (uint32_t)BX) s_loc[tid] = sched[base + min(tid, nb - 1u)] & 0x7fffffffu;
    // the batch's columns: every thread packs its ON-CHIP items' codes (an item it does not own: missing)
#pragma unroll 1
    for (int b = 0; b < BX; ++b) {
      const uint32_t loc_b = sched[base + min((uint32_t)b, nb - 1u)] & 0x7fffffffu;
      const uint32_t *col = reinterpret_cast<const uint32_t *>(p.bed + (size_t)loc_b * p.colstride);
      uint32_t word[RQ];
#pragma unroll
      for (int t = 0; t < RQ; ++t) word[t] = col[item_or_last((uint32_t)t) / 16u];
      uint32_t out[2] = {0u, 0u};
#pragma unroll
      for (int t = 0; t < RQ; ++t) {
        const uint32_t i = item_or_last((uint32_t)t);
        const uint32_t c = ((uint32_t)t < cnt && (uint32_t)b < nb) ? (word[t] >> (2u * (i % 16u))) & 3u : 1u;
        out[t / 8] |= code_nibble(c) << (4u * (uint32_t)(t % 8));
      }
      s_codes[b][tid] = make_uint2(out[0], out[1]);
    }
    uint32_t active = nb >= 32u ? 0xffffffffu : (1u << nb) - 1u;  // (uniform) locations of the batch whose inner loop still runs
    uint32_t pass = 0u;
    __syncthreads();
    TSAMD_BK(tk_setup);
    while (active != 0u) {
      pass += 1u;
      fresh();
      asm volatile("" : "+v"(vb), "+v"(vj));
      const bool vact = vmine && ((active >> vb) & 1u) != 0u;
      if (vact) eb_ran = eb_used;
#pragma unroll 1
      for (int sub = 0; sub < NSUB; ++sub) {
        if (((active >> (uint32_t)(sub * BA)) & ((1u << BA) - 1u)) == 0u) continue;  // (uniform) nobody of this sub-batch runs any more
        fresh();
        double acc0[BA][KT], acc1[BA][KT], b0[BA][BS ? KT : 1], b1[BA][BS ? KT : 1];
        uint2 cd[BA];
        const uint32_t *colp[BA];  // (uniform: the sub-batch's columns, for the streamed items' codes)
#pragma unroll
        for (int bb = 0; bb < BA; ++bb) {
          cd[bb] = s_codes[sub * BA + bb][tid];
          colp[bb] = reinterpret_cast<const uint32_t *>(p.bed + (size_t)__builtin_amdgcn_readfirstlane(s_loc[sub * BA + bb]) * p.colstride);
#pragma unroll
          for (int k = 0; k < KT; ++k) {
            acc0[bb][k] = acc1[bb][k] = 0.0;
            if constexpr (BS) {
              b0[bb][k] = s_eb[sub * BA + bb][2 * k];
              b1[bb][k] = s_eb[sub * BA + bb][2 * k + 1];
            }
          }
        }
        // one individual's weights against the sub-batch's locations (res_consume_md: the one-location path's instructions)
        auto consume = [&](const double (&wcur)[KT], const uint32_t (&nib)[BA]) {
          uint32_t zo = 0u;  // (opaque zero: the pairs are re-read from LDS per item instead of held in 4 K BA registers)
          if constexpr (!BS) asm volatile("" : "+v"(zo));
#pragma unroll
          for (int bb = 0; bb < BA; ++bb)
            res_consume_md<KT, BS>(wcur, (double)(nib[bb] & 3u), (double)(nib[bb] >> 2), b0[bb], b1[bb],
                                   reinterpret_cast<const double2 *>(&s_eb[sub * BA + bb][0]) + zo, acc0[bb], acc1[bb]);
        };
        // a streamed item's rows and its words of the sub-batch's columns (clamped: static load counts)
        auto load_streamed = [&](uint32_t s, double (&wv)[KT], uint32_t (&word)[BA]) {
          const uint32_t i = item_or_last((uint32_t)RQ + s);
#pragma unroll
          for (int k = 0; k < KT; ++k) wv[k] = (w_a + (size_t)k * np)[i];
#pragma unroll
          for (int bb = 0; bb < BA; ++bb) word[bb] = colp[bb][i / 16u];
        };
        auto streamed_codes = [&](uint32_t s, const uint32_t (&word)[BA], uint32_t (&nib)[BA]) {
          const uint32_t i = item_or_last((uint32_t)RQ + s);
#pragma unroll
          for (int bb = 0; bb < BA; ++bb)
            nib[bb] = code_nibble(((uint32_t)RQ + s < cnt && (uint32_t)(sub * BA + bb) < nb) ? (word[bb] >> (2u * (i % 16u))) & 3u : 1u);
        };
        double sa[KT], sbuf[KT];
        uint32_t worda[BA], wordb[BA];
        // the sub-batch's first streamed item is requested before its on-chip items are swept: it arrives under their arithmetic
        if (scnt_wg > 0u) load_streamed(0u, sa, worda);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < R; ++t) {
          if ((uint32_t)t >= cnt_wg) continue;
          double wcur[KT];
#pragma unroll
          for (int k = 0; k < KT; ++k) wcur[k] = buf[t][k];
          uint32_t nib[BA];
#pragma unroll
          for (int bb = 0; bb < BA; ++bb) nib[bb] = nibble_of(cd[bb], t);
          consume(wcur, nib);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          if ((uint32_t)(R + q) >= cnt_wg) continue;
          fresh();
          double wcur[KT];
#pragma unroll
          for (int k = 0; k < KT; ++k) wcur[k] = s_w[q][k][tid];
          uint32_t nib[BA];
#pragma unroll
          for (int bb = 0; bb < BA; ++bb) nib[bb] = nibble_of(cd[bb], R + q);
          consume(wcur, nib);
          __builtin_amdgcn_sched_barrier(0);
        }
        // streamed items, in index order, through two buffers, one item ahead.  The requests are UNCONDITIONAL (past the end:
        // the last item again, from the L2): behind a branch the compiler can only wait for ALL outstanding loads where the
        // paths join (s_waitcnt vmcnt(0): the request just issued included), and every second item's load stopped overlapping
        // the arithmetic (round 5: 1.5 -> 1.2 us per streamed item)
#pragma unroll 1
        for (uint32_t s = 0; s < scnt_wg; s += 2u) {
          fresh();
          uint32_t nib[BA];
          load_streamed(s + 1u, sbuf, wordb);
          __builtin_amdgcn_sched_barrier(0);
          streamed_codes(s, worda, nib);
          consume(sa, nib);
          __builtin_amdgcn_sched_barrier(0);
          if (s + 1u >= scnt_wg) break;
          fresh();
          load_streamed(s + 2u, sa, worda);
          __builtin_amdgcn_sched_barrier(0);
          streamed_codes(s + 1u, wordb, nib);
          consume(sbuf, nib);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int bb = 0; bb < BA; ++bb) res_fold<KT>(acc0[bb], acc1[bb], &s_red[sub * BA + bb][0][0], tid);
      }
      __syncthreads();
      TSAMD_BK(tk_sweep);
      fresh();
      double mine = 0.0;
      if (tid < JX) {
        mine = s_red[vb][0][vj];
#pragma unroll
        for (int wv = 1; wv < kWaves; ++wv) mine += s_red[vb][wv][vj];
      }
      xcount += 1u;
      const uint32_t tag = xseq0 + xcount;
      if (!res_exchange<KX, WR, kResOneLevelGrid, Wide>(xb, p, tag, 1u, mine, g, m, gridDim.x, s_tot, s_alive, tid, fail_code(tag, false, par, serial),
                                                       kResWaitTicks))
        return;
      TSAMD_BK(tk_xchg);
      // the BX epilogues, one lane per (location, value); the pair sum comes from the neighbouring lane (J is even)
      if (vact) {
        double nw, ebn, df;
        epilogue_values_reg(p, vj, s_tot[tid], eb_used, lam_old, nw, ebn, df);
        lam_old = nw;
        eb_used = ebn;
        s_eb[vb][vj] = ebn;
        s_diff[vb][vj] = df;
      }
      __syncthreads();
      // every wave decides for itself, lane b for location b: mean |dlambda| in the reference's order (abs_mean)
      const uint32_t lane = tid & 63u;
      bool fin = false;
      if (lane < nb && ((active >> lane) & 1u) != 0u) fin = epilogue_complete(p, pass, J, &s_diff[lane][0]);
      const uint32_t done = (uint32_t)__ballot(fin) & active;
      if (vmine && ((done >> vb) & 1u) != 0u) {
        it_mine = pass;
        if (blockIdx.x == 0) {  // the location is complete: workgroup 0 publishes its final lambda / exp(Elogbeta)
          p.lam[(size_t)vloc * J + vj] = lam_old;
          p.eb[(size_t)vloc * J + vj] = eb_used;
        }
      }
      if (blockIdx.x == 0 && tid == 0)
        for (uint32_t b = 0; b < nb; ++b)
          if ((done >> b) & 1u) count_snp_deferred(pass);
      active &= ~done;
      TSAMD_BK(tk_epi);
#ifdef TSAMD_SCHED_TIME
      tk_passes += 1u;
#endif
    }
    // the schedule's last location leaves the State the next call starts from (as ts_hybrid: its final lambda, the
    // exp(Elogbeta) its LAST executed pass used; validation mode: no gamma step will follow)
    if (base + (uint32_t)BX >= n_sched && blockIdx.x == 0) {
      if (vmine && vb == nb - 1u) {
        W->lam[vj] = lam_old;
        W->eb[vj] = eb_ran;
        if (vj == 0u) {
          W->idx = sidx + n_sched;
          W->valid = 1u;
          W->loc = vloc;
          W->hol = 1u;
          W->iters = it_mine;
          W->done = 1u;
          W->nrows = 0u;
          W->epoch = epoch_now;
          __hip_atomic_store(&ctl->last_iters, it_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (p.host_error) __hip_atomic_store(p.host_error + 1, (unsigned long long)it_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    __syncthreads();  // (the next batch overwrites s_eb / s_codes / s_loc)
  }
  if (blockIdx.x == 0) {
    if (tid == 0) {
      ctl->xseq = xseq0 + xcount;
      ctl->total_passes = tp_run;
      if (p.host_error) __hip_atomic_store(p.host_error + 2, tp_run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the histogram atomics of this thread have landed)
#ifdef TSAMD_SCHED_TIME
      if (n_sched >= 16u)
        printf("ts_hybhol n=%u batch %d x %d, %d + %d items on chip, passes of batches %u | per location (us): setup %.2f sweeps+folds %.2f exchanges %.2f "
               "epilogues %.2f | per batch pass (us): sweeps+folds %.2f exchange %.2f epilogue %.2f | whole launch %.1f us\n", n_sched, BX, BA, R, Q,
               tk_passes, tk_setup * 0.01 / n_sched, tk_sweep * 0.01 / n_sched, tk_xchg * 0.01 / n_sched, tk_epi * 0.01 / n_sched,
               tk_sweep * 0.01 / tk_passes, tk_xchg * 0.01 / tk_passes, tk_epi * 0.01 / tk_passes, (wall_clock64() - tk_start) * 0.01);
#endif
    }
    __syncthreads();
    if (p.host_error && tid < (uint32_t)TSAMD_PASS_HIST_BINS)
      __hip_atomic_store(p.host_error + 3 + tid, __hip_atomic_load(&ctl->pass_hist[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
#undef TSAMD_BK
}

}  // namespace tsamd
