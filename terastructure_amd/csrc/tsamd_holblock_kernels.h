// ts_holblock<K, WR>: a run of VALIDATION-mode SNP updates in one launch, BX locations at a time (gfx950).
//
// The reference's validation block (compute_likelihood, src/snpsamplinge.cc:476-498; snp_likelihood,
// src/snpsamplinge.hh:322-361) calls optimize_lambda(loc) once per validation location with _hol_mode set: the workers
// then skip the gamma / Elogtheta step (PhiRunnerE::do_work, src/snpsamplinge.cc:660-668), so theta is FROZEN for the
// whole block and a location's inner loop depends on nothing but its own lambda.  Distinct locations are therefore
// independent, and this kernel runs BX of them in lockstep:
//   * the shard's weights sit in registers exactly as in ts_schedule (same launch geometry, same items per thread);
//   * a pass sweeps them once per SUB-BATCH of BA locations (K = 8: two), whose 2 K BA accumulators and exp(Elogbeta)
//     values sit in vector registers for the sweep (up to K = 24; read from LDS at every use above);
//   * ONE in-launch exchange per pass carries the rows of all BX locations (up to 256 values, WideLay), and the BX
//     K x 2 epilogues run side by side on BX * 2K lanes;
//   * a location whose inner loop has ended (converged or the pass cap) is published and leaves the batch; the batch
//     ends when all its locations have.
// Every per-location sum keeps the order of the one-location path: items of a thread in order (res_consume), the
// same halving butterfly per location (res_fold<K>), the four waves in order, the same member / group order in the
// exchange -- so lambda, exp(Elogbeta), pass counts and the State left behind equal ts_schedule's for the same
// entries BIT FOR BIT (tests/test_gpu_holblock.py).
// WR > 0: one launch per rank of a sharded run, level 2 of the exchanges across the ranks (wide rows in Xchg::res_wide).
// The host (csrc/tsamd.hip) launches it for hol-mode schedules of pairwise distinct locations on a context that runs
// ts_schedule, after the first entry of the block has gone through ts_schedule (which applies the pending gamma step
// of the last training SNP, src/snpsamplinge.cc:664-667); no gamma step is ever pending when this kernel starts.
#pragma once
#include "tsamd_resident_kernels.h"

namespace tsamd {

// locations whose accumulators AND exp(Elogbeta) a thread holds at once (4 K BA <= 64 doubles: with the pairs re-read from
// LDS per item and four locations' accumulators -- round 4's first form -- a sub-batch sweep took twice the instructions) ...
constexpr int hol_sub(int k) { return k <= 4 ? 4 : k <= 8 ? 2 : 1; }
// ... and locations per exchange: a multiple of that, at most 16, rows of at most 256 values (BX K <= 128)
constexpr int hol_batch(int k) {
  const int ba = hol_sub(k);
  int n = 128 / (k * ba);
  if (n > 16 / ba) n = 16 / ba;
  if (n < 1) n = 1;
  return n * ba;
}
constexpr uint32_t kHolChunk = 1u << 14;

template <int KT, int WR>
__global__ __launch_bounds__(256, 1) void ts_holblock(Ctl *ctl_a, const double *w_a, uint32_t npad_a, uint32_t chunk_a, uint32_t par_arg,
                                                      const uint32_t *sched, uint32_t n_sched, ResXchg *xb, uint32_t serial, const DevParams p) {
  constexpr int BLOCK = 256, kWaves = BLOCK / 64, kItems = sched_items(KT, WR);
  constexpr int BA = hol_sub(KT), BX = hol_batch(KT), NSUB = BX / BA, KX = BX * KT;
  constexpr uint32_t J = 2 * KT, JX = 2 * KX;
  constexpr bool BS = KT <= 24;  // exp(Elogbeta) of the sub-batch's locations in vector registers for the sweep
  static_assert(resident_vec(KT) == 1, "one individual per item");
  static_assert(JX <= (uint32_t)BLOCK && BX <= 16 && BX % BA == 0, "a batch's row is brought by one thread per value");
  using Wide = WideLay<KX, KT>;
  __shared__ __attribute__((aligned(16))) double s_eb[BX][J];  // exp(Elogbeta) the running pass uses, per location of the batch
  __shared__ double s_diff[BX][J];
  __shared__ double s_tot[JX > 4 * J ? JX : 4 * J];
  __shared__ double s_red[BX][kWaves][J];
  // the items' genotype factors of the batch's columns, decoded ONCE per batch and packed per thread: item t holds the
  // nibble (y, 2 - y) -- (0, 0) for a missing, held-out or unowned genotype -- in bits 4t .. 4t+3.  A sweep then spends two
  // bit-field extracts and two conversions per individual where code_weights spends ten instructions (as ts_schedule does
  // per SNP: code_nibble / res_consume_md, tsamd_resident_kernels.h).
  __shared__ uint2 s_codes[BX][BLOCK];
  __shared__ int s_alive[4];

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
  uint32_t cnt = (i0 < end) ? min((end - i0 + BLOCK - 1u) / BLOCK, (uint32_t)kItems) : 0u;
  // (as in ts_schedule: everything sits in loops around fully unrolled sweeps and exchanges; left alone, the compiler
  // hoists every address that depends only on the thread out of them -- a few hundred values, spilled -- so the values
  // they derive from are made opaque per use)
  auto fresh = [&]() { asm volatile("" : "+v"(tid), "+v"(i0), "+v"(cnt)); };
  auto item_or_last = [&](uint32_t t) { return min(i0, nitems - 1u) + min(t, max(cnt, 1u) - 1u) * BLOCK; };
  // items any thread of this workgroup owns (uniform): the bodies of the others are skipped -- adding an unowned item
  // ("missing": factors of exactly 0) leaves every accumulator's bits alone, so this equals the straight-line form
  const uint32_t cnt_wg = begin < end ? min((end - begin + BLOCK - 1u) / BLOCK, (uint32_t)kItems) : 0u;
  const uint32_t g = blockIdx.x % (uint32_t)kResGroups, m = blockIdx.x / (uint32_t)kResGroups;

  if (__hip_atomic_load(&xb->abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) return;  // (see ts_resident)
  if (n_sched == 0u) {
    if (blockIdx.x == 0) carry_state(S, W, J);
    return;
  }
  if (S->valid != 0u && S->hol == 0u) {
    // a training update's gamma step is pending: the host routes the first validation entry through ts_schedule, which
    // applies it, before it launches this kernel -- anything else is a bug there; refuse loudly instead of dropping the step
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
  // the entry exchange: empty rows, nothing modified yet.  All workgroups resident?  (ts_schedule's, in its layout; sharded --
  // WR > 0, the exchange spans the ranks -- committed by a second one, as in ts_schedule)
  if (!res_exchange<KT, WR>(xb, p, xseq0 + 1u, 1u, 0.0, g, m, gridDim.x, s_tot, s_alive, tid, fail_code(xseq0 + 1u, true, par, serial),
                            WR == 0 ? (unsigned long long)p.probe_ticks : kResWaitTicks))
    return;
  if constexpr (WR > 0) {
    xcount += 1u;
    if (!res_exchange<KT, WR>(xb, p, xseq0 + xcount, 1u, 0.0, g, m, gridDim.x, s_tot, s_alive, tid, fail_code(xseq0 + xcount, true, par, serial),
                              kResWaitTicks / 3ull))
      return;
  }
  // the shard's weights: loaded once, never modified (theta is frozen in validation mode), never written back
  double buf[kItems][KT];
#pragma unroll
  for (int t = 0; t < kItems; ++t) {
#pragma unroll
    for (int k = 0; k < KT; ++k) buf[t][k] = (w_a + (size_t)k * np)[item_or_last((uint32_t)t)];
    if (t & 1) __builtin_amdgcn_sched_barrier(0);
  }
  unsigned long long tp_run = ctl->total_passes;
  uint32_t last_it = ctl->last_iters;
#ifdef TSAMD_SCHED_TIME  // diagnostic build (tools/variant.sh UNIT=hol): where a batch's time goes, 10 ns ticks, workgroup 0
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
      // a partial batch: the rows of the locations it does not have are swept with the others of their sub-batch (their results
      // are never read) -- on finite values, not on whatever the shared array held (advisor, round 4)
      s_eb[vb][vj] = 1.0;
    }
    // the batch's columns, BA at a time: every thread packs its items' codes (an item it does not own: missing)
#pragma unroll 1
    for (int sub = 0; sub < NSUB; ++sub) {
      uint32_t word[BA][kItems];
#pragma unroll
      for (int bb = 0; bb < BA; ++bb) {
        const uint32_t b = (uint32_t)(sub * BA + bb);
        const uint32_t loc_b = sched[base + min(b, nb - 1u)] & 0x7fffffffu;
        const uint32_t *col = reinterpret_cast<const uint32_t *>(p.bed + (size_t)loc_b * p.colstride);
#pragma unroll
        for (int t = 0; t < kItems; ++t) word[bb][t] = col[item_or_last((uint32_t)t) / 16u];
      }
#pragma unroll
      for (int bb = 0; bb < BA; ++bb) {
        const uint32_t b = (uint32_t)(sub * BA + bb);
        uint32_t out[2] = {0u, 0u};
#pragma unroll
        for (int t = 0; t < kItems; ++t) {
          const uint32_t i = item_or_last((uint32_t)t);
          const uint32_t c = ((uint32_t)t < cnt && b < nb) ? (word[bb][t] >> (2u * (i % 16u))) & 3u : 1u;
          out[t / 8] |= code_nibble(c) << (4u * (uint32_t)(t % 8));
        }
        s_codes[b][tid] = make_uint2(out[0], out[1]);
      }
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
#pragma unroll
        for (int bb = 0; bb < BA; ++bb) {
          cd[bb] = s_codes[sub * BA + bb][tid];
#pragma unroll
          for (int k = 0; k < KT; ++k) {
            acc0[bb][k] = acc1[bb][k] = 0.0;
            if constexpr (BS) {
              b0[bb][k] = s_eb[sub * BA + bb][2 * k];
              b1[bb][k] = s_eb[sub * BA + bb][2 * k + 1];
            }
          }
        }
#pragma unroll
        for (int t = 0; t < kItems; ++t) {
          if ((uint32_t)t >= cnt_wg) continue;
          double wcur[KT];
#pragma unroll
          for (int k = 0; k < KT; ++k) wcur[k] = buf[t][k];
          uint32_t zo = 0u;  // (K > 24, opaque zero: exp(Elogbeta) is re-read from LDS per item instead of held in 4 K registers)
          if constexpr (!BS) asm volatile("" : "+v"(zo));
#pragma unroll
          for (int bb = 0; bb < BA; ++bb) {
            const uint32_t nib = nibble_of(cd[bb], t);
            res_consume_md<KT, BS>(wcur, (double)(nib & 3u), (double)(nib >> 2), b0[bb], b1[bb],
                                reinterpret_cast<const double2 *>(&s_eb[sub * BA + bb][0]) + zo, acc0[bb], acc1[bb]);
          }
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
    // the schedule's last location leaves the State the next call starts from (as ts_schedule: its final lambda, the
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
    __syncthreads();  // (the next batch overwrites s_eb / s_codes)
  }
  (void)last_it;
  if (blockIdx.x == 0) {
    if (tid == 0) {
      ctl->xseq = xseq0 + xcount;
      ctl->total_passes = tp_run;
      if (p.host_error) __hip_atomic_store(p.host_error + 2, tp_run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the histogram atomics of this thread have landed)
#ifdef TSAMD_SCHED_TIME
      if (n_sched >= 16u)
        printf("ts_holblock n=%u batch %d passes of batches %u | per location (us): setup %.2f sweeps+folds %.2f exchanges %.2f epilogues %.2f | "
               "per batch pass (us): sweeps+folds %.2f exchange %.2f epilogue %.2f | whole launch %.1f us\n", n_sched, BX, tk_passes,
               tk_setup * 0.01 / n_sched, tk_sweep * 0.01 / n_sched, tk_xchg * 0.01 / n_sched, tk_epi * 0.01 / n_sched, tk_sweep * 0.01 / tk_passes,
               tk_xchg * 0.01 / tk_passes, tk_epi * 0.01 / tk_passes, (wall_clock64() - tk_start) * 0.01);
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
