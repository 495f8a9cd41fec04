// The register-resident kernels (gfx950): the shard's N x K fp64 weights live in the register file.
//
//   ts_resident<K>            ALL plain passes of one SNP in one launch (launch mode "per SNP")
//   ts_schedule<K,PARTIAL,WR> a WHOLE schedule in one launch: gamma steps, passes, epilogues (mode "per schedule")
//
// 256 workgroups of 256 threads, one per compute unit, ONE wave per SIMD: a thread owns the whole 512-entry register
// file.  Per K a thread holds resident_items(K) items of one individual each:
//   K <= 8      16 items                                         -> 1 048 576 individuals per GPU
//   K = 9..16   floor(128 / K) items (14 at K = 9 ... 8 at K = 16)  -> 917 504 ... 524 288
//   K = 17..24  floor(112 / K) items (6 at K = 17 ... 4 at K = 24; K = 22: 4)  -> 393 216 ... 262 144
//   K = 25..32  3 items                                           -> 196 608
// i.e. at most 128 doubles = 256 registers of weights per thread (the AGPR half of the file) next to the gamma
// step's K-sized temporaries (which is why the budget shrinks as K grows).  Between passes the workgroups all-reduce their partial rows INSIDE the launch
// (ResXchg below).  Restated reference code: as ts_pass (tsamd_kernels.h) -- PhiRunnerE::process / update_phimom /
// update_phidad / update_lambda_t (src/snpsamplinge.hh:276-300, :416-431, src/snpsamplinge.cc:742-759),
// update_gamma / estimate_theta (:695-740), update_lambda / estimate_beta (:267-296), optimize_lambda (:320-366).
#pragma once
#include <type_traits>

#include "tsamd_kernels.h"

namespace tsamd {
#ifndef TSAMD_NOCOL_K
#define TSAMD_NOCOL_K 0
#endif
#ifndef TSAMD_GAMMA_SB_READLANE  // (experiment switch: 0 = the gamma step re-reads exp(Elogbeta) of the last pass from LDS at every use, for every K)
#define TSAMD_GAMMA_SB_READLANE 1
#endif
#ifndef TSAMD_REPL_READLANE  // (experiment switch: 0 = the per-wave form broadcasts exp(Elogbeta) through its LDS row, as in rounds 3-4)
#define TSAMD_REPL_READLANE 1
#endif

// ---- geometry per K ---------------------------------------------------------------------------------------------
constexpr int kResidentMaxK = 32;
constexpr int kResidentBlock = 256;
#ifdef TSAMD_RES_VEC  // (experiments, UNIT=all tools/variant.sh: 2 = pairs of individuals per item at K <= 8, round 2's geometry)
constexpr int resident_vec(int k) { return k <= 8 ? TSAMD_RES_VEC : 1; }
#else
constexpr int resident_vec(int) { return 1; }
#endif
// (K = 22: 4, not floor(112 / 22) = 5 -- with 110 doubles of weights every ts_schedule<22> instantiation spilled 36 ... 76 bytes
// to scratch: profiles/r06_kernel_resources.txt, round 6)
constexpr int resident_items(int k) { return k <= 8 ? 16 / resident_vec(k) : k <= 16 ? 128 / k : k == 22 ? 4 : k <= 24 ? 112 / k : 3; }
// individuals a workgroup can hold
constexpr int resident_capacity(int k) { return resident_items(k) * resident_vec(k) * kResidentBlock; }
// ... and what a thread of a SHARDED launch holds (ts_schedule<K, ., WR > 0>, ts_holblock<K, WR > 0>: the ranks' launches share one
// geometry rule, resident_geometry in csrc/tsamd.hip): K = 16 one item less -- its 128 doubles of weights fill the AGPR half of the
// register file, and the sharded exchange's few extra registers went to scratch (20 ... 52 bytes); K = 14 (9 x 14 = 126 doubles) likewise
constexpr int sharded_items(int k) { return k == 16 ? 7 : k == 14 ? 8 : resident_items(k); }
constexpr int sched_items(int k, int wr) { return wr > 0 ? sharded_items(k) : resident_items(k); }

// ---- in-launch exchange -----------------------------------------------------------------------------------------
// Per pass every workgroup contributes its partial row (2K doubles) and every workgroup gets the fixed-order total,
// as a two-level all-reduce over 8-byte granules {tag, 32-bit half of a double}, each written by ONE relaxed
// agent-scope store: the data is the flag (MI355X_MICROARCH.md, hand-off price list) -- no fence, no second trip.
//   level 1: the workgroups of group g = blockIdx % 8 (= the XCD) publish rows(g, blockIdx / 8); the group's leader
//            (blockIdx < 8) re-reads them until every tag matches, adds them in member order, publishes sums(g);
//   level 2: every workgroup re-reads the 8 group rows and adds them in group order (sharded: world x 8 rows in its
//            own rank's Xchg::res_sums, written by the leaders of all ranks over xGMI -- polled in two halves, res_split).
// A single workgroup exchanges nothing.  Up to kResOneLevelGrid = 32 workgroups on one GPU (shards up to 8 192 individuals:
// the sizes of real cohorts; K <= 8 only in ts_resident) there is only ONE level: every workgroup sweeps every row.  The layouts of the two forms overlap, which
// is harmless: an exchange is self-contained (tags never repeat).
// A row is cut into column blocks of 32 granules (16 values); wave w of a workgroup sweeps blocks w, w + 4, ...: one
// wave at K <= 8, two at K <= 16, four at K <= 32 -- the sweeps of a wide row run side by side on the four SIMDs.
// An exchange can carry TWO rows (regions A and B): B is the row of a SNP's LAST pass, deferred into the first
// exchange of the next SNP (ts_schedule), whose total only workgroup 0 needs (it publishes that SNP's lambda).
// tag = a counter in Ctl (xseq) that never repeats, so nothing is re-initialised between launches.  Every wait is
// bounded; a failure sets abort_word, which ends all later waits at once and turns every later kernel of the context
// into a no-op until the host has dealt with it (tsamd_synchronize).
constexpr int kResGroups = 8;    // (Xchg::res_sums is laid out for these two)
constexpr int kResMembers = 32;  // workgroups per group (grid <= 256)
#ifndef TSAMD_ONE_LEVEL  // (experiments: 0 = always two levels.  Measured at K = 8: up to 16 rows 34.5 us per update against
#define TSAMD_ONE_LEVEL 32  // 43.8 with two levels; 17 ... 32 rows -- since the row sums run on the vector ALU -- 33.8 against
#endif                      // 37.1 at N = 16 000; 64 loses at every size: profiles/r03_experiments.md)
constexpr int kResOneLevelGrid = TSAMD_ONE_LEVEL;  // up to this many workgroups (one GPU) the exchange has ONE level: everybody reads every row
constexpr int res_blocks(int k) { return (4 * k + 31) / 32; }  // 32-granule column blocks of a row of 2K values
constexpr int kResMaxGran = 32 * res_blocks(kResidentMaxK);
constexpr int kResRegionRows = kResGroups * kResMembers + 2 * kResGroups;  // member rows, then two slots of group sums
// ts_holblock (tsamd_holblock_kernels.h) exchanges the rows of a whole BATCH of validation locations at once: up to 256
// values = kResWideGran granules per row, in an area of its own (a row of the wide layout would land on the narrow
// layout's group sums, which a slow workgroup may still be polling for the launch's entry exchange).
constexpr int kResWideGran = 512;
struct ResXchg {
  unsigned long long abort_word;  // a bounded wait gave up (its tag); every later wait and kernel returns at once
  unsigned long long pad_[31];
  unsigned long long gran[2 * kResRegionRows * kResMaxGran];  // regions A and B, laid out for the context's K (ResLay)
  unsigned long long wide[kResRegionRows * kResWideGran];     // one region of wide rows (WideLay)
};
template <int KT>
struct ResLay {
  static constexpr int kSegK = KT;  // (see WideLay)
  static constexpr uint32_t GR = 32u * (uint32_t)res_blocks(KT);  // granules per row (row stride)
  static constexpr uint32_t kRegion = (uint32_t)kResRegionRows * GR;
  static __device__ __forceinline__ unsigned long long *rows(ResXchg *xb, uint32_t region, uint32_t g, uint32_t m) {
    return xb->gran + region * kRegion + (g * (uint32_t)kResMembers + m) * GR;
  }
  static __device__ __forceinline__ unsigned long long *sums(ResXchg *xb, uint32_t region, uint32_t slot, uint32_t g) {
    return xb->gran + region * kRegion + ((uint32_t)(kResGroups * kResMembers) + slot * (uint32_t)kResGroups + g) * GR;
  }
  // one level (grid <= kResOneLevelGrid): row `row` of slot `slot`, laid over the member rows
  static __device__ __forceinline__ unsigned long long *flat(ResXchg *xb, uint32_t region, uint32_t slot, uint32_t row) {
    return xb->gran + region * kRegion + (slot * (uint32_t)kResOneLevelGrid + row) * GR;
  }
  // sharded: level 2 in Xchg::res_sums of every rank, row r * 8 + g of (slot, region)
  static __device__ __forceinline__ unsigned long long *rank_sums(Xchg *x, uint32_t region, uint32_t slot, uint32_t row) {
    return x->res_sums + ((slot * 2u + region) * (uint32_t)(kMaxRanks * kResGroups) + row) * GR;
  }
  // sharded, three levels, rows of at most two column blocks (res_split): the SECOND half's partial sum of a leader's level-2
  // poll (the first goes to sums()), in the part of ResXchg::gran that rows of at most 64 granules leave unused
  static __device__ __forceinline__ unsigned long long *sums_hi(ResXchg *xb, uint32_t region, uint32_t slot, uint32_t g) {
    static_assert(GR > 64u || 2u * kRegion + 2u * 2u * (uint32_t)kResGroups * GR <= 2u * (uint32_t)kResRegionRows * (uint32_t)kResMaxGran,
                  "the second halves' rows lie behind regions A and B");
    return xb->gran + 2u * kRegion + ((region * 2u + slot) * (uint32_t)kResGroups + g) * GR;
  }
};
// Sharded launches (WR > 0) whose rows have at most two column blocks (K <= 16) poll the world x 8 rows of level 2 in TWO
// HALVES ON TWO WAVES side by side (round 6): a lane then holds WR / 2 row pairs in flight instead of WR (each an 8-byte
// load, two registers) -- what the 8-rank instantiations of K <= 8 spilled to scratch for -- at the same single memory
// round trip.  The halves' partial sums land in s_tot[i] and s_tot[4K + i]; res_total adds them (half 0 + half 1, the
// same order on every rank).  Everything else: s_tot[i] is the total.
template <int KT, int WR, class LAY = ResLay<KT>>
constexpr bool res_split() {
  return WR > 0 && res_blocks(KT) <= 2 && std::is_same<LAY, ResLay<KT>>::value;
}
template <int KT, int WR>
__device__ __forceinline__ double res_total(const double *s_tot, uint32_t i) {
  if constexpr (res_split<KT, WR>()) return s_tot[i] + s_tot[4u * (uint32_t)KT + i]; else return s_tot[i];
}
// 5 ... 8 ranks (WR = 32), the K whose instantiation would otherwise still spill to scratch (profiles/r06_kernel_resources.txt):
//   K = 18 (rows of three column blocks: every wave polls a block of its own, 32 row pairs per lane in one go at the other K >= 17)
//          polls its 32 pairs as 16 + 16, one after the other;
//   K = 16 (split as above: 16 pairs per lane and wave) polls them as 8 + 8
// -- one more memory round trip per exchange, added in the same order as in one go.
constexpr bool res_seq_halves(int k) { return k == 18 || k == 16; }
// doubles of s_tot a kernel with narrow rows of 2 KT values provides
template <int KT, int WR>
constexpr int res_tot_doubles() { return (res_split<KT, WR>() ? 8 : 4) * KT; }

// the same row / sum / flat positions for rows of 2 KX values in ResXchg::wide (one region)
// KS: the K of the context (rows of 2 KS values in its training kernels): a sharded exchange adds the ranks' rows in the order
// those kernels do (kSegmented / res_segment in res_exchange), so that a batched validation block equals the entry-by-entry path
// bit for bit
template <int KX, int KS = KX>
struct WideLay {
  static constexpr int kSegK = KS;
  static constexpr uint32_t GR = 32u * (uint32_t)res_blocks(KX);
  static_assert(GR <= (uint32_t)kResWideGran, "a wide row holds at most kResWideGran granules");
  static __device__ __forceinline__ unsigned long long *rows(ResXchg *xb, uint32_t, uint32_t g, uint32_t m) {
    return xb->wide + (g * (uint32_t)kResMembers + m) * GR;
  }
  static __device__ __forceinline__ unsigned long long *sums(ResXchg *xb, uint32_t, uint32_t slot, uint32_t g) {
    return xb->wide + ((uint32_t)(kResGroups * kResMembers) + slot * (uint32_t)kResGroups + g) * GR;
  }
  static __device__ __forceinline__ unsigned long long *flat(ResXchg *xb, uint32_t, uint32_t slot, uint32_t row) {
    return xb->wide + (slot * (uint32_t)kResOneLevelGrid + row) * GR;
  }
  // sharded: level 2 in Xchg::res_wide of every rank, row r * 8 + g of the slot
  static __device__ __forceinline__ unsigned long long *rank_sums(Xchg *x, uint32_t, uint32_t slot, uint32_t row) {
    return x->res_wide + (slot * (uint32_t)(kMaxRanks * kResGroups) + row) * GR;
  }
};
static_assert(sizeof(((Xchg *)nullptr)->res_wide) / sizeof(unsigned long long) == 2u * kMaxRanks * kResGroups * (unsigned)kResWideGran,
              "Xchg::res_wide holds two slots of world x 8 wide rows");

constexpr unsigned long long kResWaitTicks = 300000000ull;  // 3 s at 100 MHz
// what a failing wait leaves in the pinned host word [0] (DevParams::host_error): the tag, whether the launch had
// modified anything yet (bit 32: no -- the failure hit its entry exchange; the state the launch started from is intact),
// the launch parity (bit 33) and the host's launch serial (bits 34 ...)
constexpr unsigned long long kFailIntact = 1ull << 32;
// ... and a wait that gives up AFTER its launch has modified something also leaves its code in a second pinned word
// (DevParams::host_error + kHostDirtyWord).  The entry exchange is one bounded wait per workgroup, and workgroups can
// disagree on its outcome (the last row arrives right at the deadline: one workgroup gives up "intact", the others pass,
// start modifying state and only notice the abort word at their next exchange): the host replays from the state the
// launch started from only when this word is still clear.
constexpr int kHostDirtyWord = 3 + TSAMD_PASS_HIST_BINS;
__device__ __forceinline__ unsigned long long fail_code(uint32_t tag, bool intact, uint32_t par, uint32_t serial) {
  return (unsigned long long)tag | (intact ? kFailIntact : 0ull) | ((unsigned long long)(par & 1u) << 33) | ((unsigned long long)serial << 34);
}

// One wave sweeps N row pairs of one column block: lane l, load i = granule col0 + l % 32 of row 2 i + (l >= 32);
// rows >= row_limit do not exist, granules >= nvalid carry nothing.  Polls until every existing granule carries
// `tag`; false when the bounded wait gave up (abort_word set, host told).
template <int N, int SCOPE>
__device__ __forceinline__ bool res_sweep(const unsigned long long *base, uint32_t stride, uint32_t col0, uint32_t tag, uint32_t nvalid,
                                          uint32_t row_limit, unsigned (&v)[N], unsigned long long *abort_word,
                                          unsigned long long *host_flag, unsigned long long code, unsigned long long ticks, uint32_t lane) {
  const uint32_t c = lane & 31u;
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const uint32_t row = 2u * (uint32_t)i + (lane >> 5);
      const bool exists = c < nvalid && row < row_limit;
      const unsigned long long x = __hip_atomic_load(base + (size_t)row * stride + col0 + c, __ATOMIC_RELAXED, SCOPE);
      v[i] = exists ? (unsigned)x : 0u;
      ok &= !exists || (unsigned)(x >> 32) == tag;
    }
    if (__all(ok)) return true;
    if (wall_clock64() - t0 > ticks || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) {
      if (lane == 0) {
        __hip_atomic_store(abort_word, (unsigned long long)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (host_flag) {
          if ((code & kFailIntact) == 0ull) __hip_atomic_store(host_flag + kHostDirtyWord, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(host_flag, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// The same for NB column blocks of the same rows in ONE poll (wide rows, ts_holblock: a wave sweeps blocks first, first + 4,
// ... of a row of gran_total granules): the blocks of a row are posted together, so one memory round trip serves them all
// where NB consecutive res_sweep calls take NB.  Blocks past nblk do not exist (their loads are clamped and ignored).
template <int N, int NB, int SCOPE>
__device__ __forceinline__ bool res_sweep_blocks(const unsigned long long *base, uint32_t stride, uint32_t first, uint32_t nblk, uint32_t gran_total,
                                                 uint32_t tag, uint32_t row_limit, unsigned (&v)[NB][N], unsigned long long *abort_word,
                                                 unsigned long long *host_flag, unsigned long long code, unsigned long long ticks, uint32_t lane) {
  const uint32_t c = lane & 31u;
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const uint32_t q = first + 4u * (uint32_t)u, qc = min(q, nblk - 1u);
      const uint32_t nvalid = q < nblk ? min(32u, gran_total - 32u * q) : 0u;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const uint32_t row = 2u * (uint32_t)i + (lane >> 5);
        const bool exists = c < nvalid && row < row_limit;
        const unsigned long long x = __hip_atomic_load(base + (size_t)row * stride + 32u * qc + c, __ATOMIC_RELAXED, SCOPE);
        v[u][i] = exists ? (unsigned)x : 0u;
        ok &= !exists || (unsigned)(x >> 32) == tag;
      }
    }
    if (__all(ok)) return true;
    if (wall_clock64() - t0 > ticks || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) {
      if (lane == 0) {
        __hip_atomic_store(abort_word, (unsigned long long)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (host_flag) {
          if ((code & kFailIntact) == 0ull) __hip_atomic_store(host_flag + kHostDirtyWord, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(host_flag, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// sum of the N row pairs a wave has swept (lo / hi halves of a double sit in neighbouring lanes): on return lane 2 j
// (j < 16) holds the total of value j of the block over rows 0, 2, 4, ... plus rows 1, 3, 5, ...  All on the vector
// ALU: the partner's half comes over DPP (quad_perm), the other half-wave's sum over v_permlane32_swap -- through the
// LDS crossbar (ds_bpermute, what __shfl_xor compiles to) the N dependent round trips of a leader's sweep alone cost
// about as much as a hop of the exchange.  An even lane holds the low half and receives the high one; what the odd
// lanes assemble from the same two words is not a number anybody reads.
template <int N>
__device__ __forceinline__ double res_rows_add(const unsigned (&v)[N], double s) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const unsigned other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v[i], 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false);
    s += __longlong_as_double(((unsigned long long)other << 32) | v[i]);
  }
  return s;
}
template <int N>
__device__ __forceinline__ double res_sum(const unsigned (&v)[N], uint32_t lane) {
  (void)lane;
  const double s = res_rows_add<N>(v, 0.0);
  return pair_add<32>(s, s);  // own + lane ^ 32
}

// rows [A, B) of v added onto s in order
template <int A, int B, int N>
__device__ __forceinline__ double res_rows_add_range(const unsigned (&v)[N], double s) {
  static_assert(A >= 0 && A <= B && B <= N, "range");
#pragma unroll
  for (int i = A; i < B; ++i) {
    const unsigned other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v[i], 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false);
    s += __longlong_as_double(((unsigned long long)other << 32) | v[i]);
  }
  return s;
}
// total of one SEGMENT of row pairs [A, B): (its even rows) + (its odd rows)
template <int A, int B, int N>
__device__ __forceinline__ double res_segment(const unsigned (&v)[N]) {
  const double s = res_rows_add_range<A, B, N>(v, 0.0);
  return pair_add<32>(s, s);
}

__device__ __forceinline__ void res_post(unsigned long long *dst, uint32_t tag, double v, int scope_system) {
  const unsigned long long bits = __double_as_longlong(v);
  const unsigned long long lo = ((unsigned long long)tag << 32) | (uint32_t)bits, hi = ((unsigned long long)tag << 32) | (uint32_t)(bits >> 32);
  if (scope_system) {
    __hip_atomic_store(dst, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dst + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    __hip_atomic_store(dst, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The exchange, called by ALL threads of a workgroup after a workgroup barrier.  Thread tid < J brings value tid of the
// region-A row in `mine`; with width == 2 thread 64 + j brings value j of the region-B row (rows of up to 64 values;
// wider rows -- LAY = WideLay -- have one region only).  On return (after a
// workgroup barrier) s_tot[0][j] holds the region-A totals in every workgroup and s_tot[1][j] the region-B totals in
// workgroup 0.  WR: row pairs per lane of the cross-rank level 2 (0: one GPU).  false: a bounded wait gave up.
// `overlap` is called exactly once by every thread, at the point where the workgroup has nothing to do but wait: a member
// right after it has posted its row, a leader after it has published its group's sum (its level-1 sweep is on everybody's
// critical path).  ts_hybrid requests the next pass' first streamed items there: the loads travel while the sums do.
struct NoOverlap {
  __device__ __forceinline__ void operator()() const {}
};
template <int KT, int WR, int ONE = kResOneLevelGrid, class LAY = ResLay<KT>, class OV = NoOverlap>
__device__ __forceinline__ bool res_exchange(ResXchg *xb, const DevParams &p, uint32_t tag, uint32_t width, double mine, uint32_t g, uint32_t m,
                                             uint32_t grid, double *s_tot /* [2][2K] */, int *s_alive /* [4], all 1 */, uint32_t tid,
                                             unsigned long long code, unsigned long long ticks, OV overlap = OV()) {
  using L = LAY;
  constexpr uint32_t J = 2 * KT, RB = (uint32_t)res_blocks(KT), GR = L::GR;
  constexpr int kPerWave = (2 * (int)RB + 3) / 4;  // column blocks a wave sweeps at most
  constexpr bool kWideRow = J > 64u;  // a row wider than a wave (ts_holblock): ONE region, thread tid brings value tid
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  const uint32_t nblk = width * RB;
  // a wave that has swept column block cb of `region` holds the total of value 16 cb + j' in lane 2 j' (j' < nvalid / 2)
  auto sink = [&](uint32_t region, uint32_t cb, uint32_t nvalid, uint32_t ln, double sv) {
    if (ln < nvalid && !(ln & 1u)) s_tot[region * J + 16u * cb + (ln >> 1)] = sv;
  };
  if constexpr (WR == 0) {
    if (grid == 1u) {  // ONE workgroup (the smallest cohorts): its row is the total, nothing goes through memory
      const uint32_t region = kWideRow ? 0u : tid >> 6, j = kWideRow ? tid : tid & 63u;
      if (region < width && j < J) s_tot[region * J + j] = mine;
      overlap();
      __syncthreads();
      return true;
    }
  }
  if constexpr (WR == 0 && ONE > 0) {
    if (grid <= (uint32_t)ONE) {
      // Few workgroups (shards up to 4 096 individuals: the sizes of real cohorts): ONE level.  Every workgroup posts its
      // row(s) into slot tag & 1 of a flat array (row = blockIdx) and sweeps all rows itself, adding them in workgroup
      // order: one store -> visible -> load chain instead of two.  Two slots: nobody can post exchange x + 2 before
      // everybody has posted x + 1, i.e. has finished reading x.
      const uint32_t slot = tag & 1u;
      {
        const uint32_t region = kWideRow ? 0u : tid >> 6, j = kWideRow ? tid : tid & 63u;
        if (region < width && j < J) res_post(L::flat(xb, region, slot, blockIdx.x) + 2u * j, tag, mine, 0);
      }
      overlap();
      bool alive1 = true;
      if constexpr (kWideRow) {
        // a wide row (ts_holblock): all of this wave's column blocks in one poll
        constexpr int NBW = ((int)RB + 3) / 4;
        auto sweep_all = [&](auto n_c) {
          constexpr int N = decltype(n_c)::value;
          constexpr int NBP = N > 8 ? 2 : 4;  // blocks per poll (16 row pairs x 2 blocks = 64 loads in flight per lane)
#pragma unroll
          for (int u0 = 0; u0 < NBW; u0 += NBP) {
            if (wave + 4u * (uint32_t)u0 >= RB) break;
            unsigned v[NBP][N];
            alive1 = res_sweep_blocks<N, NBP, __HIP_MEMORY_SCOPE_AGENT>(L::flat(xb, 0u, slot, 0), GR, wave + 4u * (uint32_t)u0, RB, 2u * J, tag, grid, v,
                                                                       &xb->abort_word, p.host_error, code, ticks, lane) && alive1;
#pragma unroll
            for (int u = 0; u < NBP; ++u) {
              const uint32_t q = wave + 4u * (uint32_t)(u0 + u);
              if (q < RB) {
                const uint32_t nvalid = min(32u, 2u * J - 32u * q);
                const double sv = res_sum<N>(v[u], lane);
                if (lane < nvalid && !(lane & 1u)) s_tot[16u * q + (lane >> 1)] = sv;
              }
            }
          }
        };
        if (ONE <= 16 || grid <= 16u) sweep_all(std::integral_constant<int, 8>{}); else sweep_all(std::integral_constant<int, (ONE > 16 ? ONE / 2 : 8)>{});
      } else {
#pragma unroll
      for (int u = 0; u < kPerWave; ++u) {
        const uint32_t q = wave + 4u * (uint32_t)u;
        if (q < nblk && (q < RB || blockIdx.x == 0)) {
          const uint32_t region = q / RB, cb = q % RB, nvalid = min(32u, 2u * J - 32u * cb);
          double s = 0.0;
          if (ONE <= 16 || grid <= 16u) {
            unsigned v[8];
            alive1 = res_sweep<8, __HIP_MEMORY_SCOPE_AGENT>(L::flat(xb, region, slot, 0), GR, 32u * cb, tag, nvalid, grid, v, &xb->abort_word, p.host_error,
                                                            code, ticks, lane) && alive1;
            s = res_sum<8>(v, lane);
          } else if constexpr (ONE > 16) {
            unsigned v[ONE / 2];
            alive1 = res_sweep<ONE / 2, __HIP_MEMORY_SCOPE_AGENT>(L::flat(xb, region, slot, 0), GR, 32u * cb, tag, nvalid, grid, v, &xb->abort_word,
                                                                  p.host_error, code, ticks, lane) && alive1;
            s = res_sum<ONE / 2>(v, lane);
          }
          sink(region, cb, nvalid, lane, s);
        }
      }
      }
      if (lane == 0 && !alive1) s_alive[wave] = 0;
      __syncthreads();
      return (s_alive[0] & s_alive[1] & s_alive[2] & s_alive[3]) != 0;
    }
  }
  {
    const uint32_t region = kWideRow ? 0u : tid >> 6, j = kWideRow ? tid : tid & 63u;
    if (region < width && j < J) res_post(L::rows(xb, region, g, m) + 2u * j, tag, mine, 0);
  }
  if (m != 0u) overlap();
  const uint32_t members = g < grid ? (grid - g + (uint32_t)kResGroups - 1u) / (uint32_t)kResGroups : 0u;
  const uint32_t groups = min(grid, (uint32_t)kResGroups);
  bool alive = true;
  // level 1 (leaders): all of this wave's blocks first, so that every group sum is on its way before anybody waits for one
  if (m == 0) {
    if constexpr (kWideRow && WR == 0) {
      constexpr int NBW = ((int)RB + 3) / 4, NBP = 2;  // two blocks per poll: 16 row pairs x 2 = 64 loads in flight per lane
#pragma unroll
      for (int u0 = 0; u0 < NBW; u0 += NBP) {
        if (wave + 4u * (uint32_t)u0 >= RB) break;
        unsigned v[NBP][kResMembers / 2];
        alive = res_sweep_blocks<kResMembers / 2, NBP, __HIP_MEMORY_SCOPE_AGENT>(L::rows(xb, 0u, g, 0), GR, wave + 4u * (uint32_t)u0, RB, 2u * J, tag, members,
                                                                                v, &xb->abort_word, p.host_error, code, ticks, lane) && alive;
#pragma unroll
        for (int u = 0; u < NBP; ++u) {
          const uint32_t q = wave + 4u * (uint32_t)(u0 + u);
          if (q < RB) {
            const uint32_t nvalid = min(32u, 2u * J - 32u * q);
            const double sv = res_sum<kResMembers / 2>(v[u], lane);
            if (lane < nvalid && !(lane & 1u)) res_post(L::sums(xb, 0u, tag & 1u, g) + 32u * q + lane, tag, sv, 0);
          }
        }
      }
    } else {
#pragma unroll
    for (int u = 0; u < kPerWave; ++u) {
      const uint32_t q = wave + 4u * (uint32_t)u;
      if (q < nblk) {
        const uint32_t region = q / RB, cb = q % RB, nvalid = min(32u, 2u * J - 32u * cb);
        unsigned v[kResMembers / 2];
        alive = res_sweep<kResMembers / 2, __HIP_MEMORY_SCOPE_AGENT>(L::rows(xb, region, g, 0), GR, 32u * cb, tag, nvalid, members, v, &xb->abort_word,
                                                                     p.host_error, code, ticks, lane) && alive;
        const double s = res_sum<kResMembers / 2>(v, lane);
        if (lane < nvalid && !(lane & 1u)) {
          if constexpr (WR == 0) {
            res_post(L::sums(xb, region, tag & 1u, g) + 32u * cb + lane, tag, s, 0);
          } else {  // sharded: the group sum goes to every rank (this one included), straight over xGMI
            for (uint32_t r = 0; r < p.xchg_world; ++r)
              res_post(L::rank_sums(p.peers[r], region, tag & 1u, p.xchg_rank * (uint32_t)kResGroups + g) + 32u * cb + lane, tag, s, 1);
          }
        }
      }
    }
    }
    overlap();
  }
  // level 2: region A everybody, region B workgroup 0 only
  if constexpr (kWideRow && WR == 0) {
    constexpr int NBW = ((int)RB + 3) / 4;
    unsigned v2[NBW][kResGroups / 2];
    alive = res_sweep_blocks<kResGroups / 2, NBW, __HIP_MEMORY_SCOPE_AGENT>(L::sums(xb, 0u, tag & 1u, 0), GR, wave, RB, 2u * J, tag, groups, v2,
                                                                           &xb->abort_word, p.host_error, code, ticks, lane) && alive;
#pragma unroll
    for (int u = 0; u < NBW; ++u) {
      const uint32_t q = wave + 4u * (uint32_t)u;
      if (q < RB) {
        const uint32_t nvalid = min(32u, 2u * J - 32u * q);
        const double sv = res_sum<kResGroups / 2>(v2[u], lane);
        if (lane < nvalid && !(lane & 1u)) s_tot[16u * q + (lane >> 1)] = sv;
      }
    }
  } else if constexpr (res_split<KT, WR, LAY>()) {
    // sharded, rows of one or two column blocks: unit (block q, half h) = wave, wave + 4, ...; half h are rows
    // [h WR, (h + 1) WR) of the world x 8 -- two waves poll a block side by side, WR / 2 row pairs per lane each
    constexpr int H = WR / 2, kUnitsPerWave = (4 * (int)RB + 3) / 4;
    const uint32_t rows_all = p.xchg_world * (uint32_t)kResGroups;
#pragma unroll
    for (int u = 0; u < kUnitsPerWave; ++u) {
      const uint32_t unit = wave + 4u * (uint32_t)u, q = unit >> 1, h = unit & 1u;
      if (q < nblk && (q < RB || blockIdx.x == 0)) {
        const uint32_t region = q / RB, cb = q % RB, nvalid = min(32u, 2u * J - 32u * cb), row0 = h * (uint32_t)WR;
        const uint32_t at = region * J + 16u * cb + (lane >> 1);
        if (p.xchg_gather_leaders == 0u || m == 0u) {
          double s;
          if constexpr (WR == 32 && res_seq_halves(KT)) {  // (16 row pairs per lane are still too many for this instantiation: 8 + 8)
            const uint32_t rows_h = rows_all > row0 ? rows_all - row0 : 0u;
            unsigned va[H / 2];
            alive = res_sweep<H / 2, __HIP_MEMORY_SCOPE_SYSTEM>(L::rank_sums(p.xchg, region, tag & 1u, row0), GR, 32u * cb, tag, nvalid, min(rows_h, (uint32_t)H), va,
                                                                &xb->abort_word, p.host_error, code, ticks, lane) && alive;
            double acc = res_rows_add<H / 2>(va, 0.0);
            if (rows_h > (uint32_t)H) {  // (uniform)
              unsigned vb[H / 2];
              alive = res_sweep<H / 2, __HIP_MEMORY_SCOPE_SYSTEM>(L::rank_sums(p.xchg, region, tag & 1u, row0 + (uint32_t)H), GR, 32u * cb, tag, nvalid,
                                                                  rows_h - (uint32_t)H, vb, &xb->abort_word, p.host_error, code, ticks, lane) && alive;
              acc = res_rows_add<H / 2>(vb, acc);
            }
            s = pair_add<32>(acc, acc);
          } else {
          unsigned v2[H];
          alive = res_sweep<H, __HIP_MEMORY_SCOPE_SYSTEM>(L::rank_sums(p.xchg, region, tag & 1u, row0), GR, 32u * cb, tag, nvalid,
                                                          rows_all > row0 ? rows_all - row0 : 0u, v2, &xb->abort_word, p.host_error, code, ticks, lane) && alive;
          s = res_sum<H>(v2, lane);
          }
          if (lane < nvalid && !(lane & 1u)) {
            s_tot[h * 2u * J + at] = s;
            // three levels (TSAMD_SCHEDULE_GATHER=leaders): the leader hands BOTH partial sums to the members of its group
            if (p.xchg_gather_leaders != 0u)
              res_post((h ? L::sums_hi(xb, region, tag & 1u, g) : L::sums(xb, region, tag & 1u, g)) + 32u * cb + lane, tag, s, 0);
          }
        } else if (h == 0u) {
          // a member: its leader's two partial sums in one poll (lanes < 32 the first half's row, the others the second's);
          // res_sum<1> adds them in the order res_total does
          unsigned v1[1];
          unsigned long long *lo = L::sums(xb, region, tag & 1u, g);
          alive = res_sweep<1, __HIP_MEMORY_SCOPE_AGENT>(lo, (uint32_t)(L::sums_hi(xb, region, tag & 1u, g) - lo), 32u * cb, tag, nvalid, 2u, v1, &xb->abort_word,
                                                         p.host_error, code, ticks, lane) && alive;
          const double s = res_sum<1>(v1, lane);
          if (lane < nvalid && !(lane & 1u)) {
            s_tot[at] = s;
            s_tot[2u * J + at] = 0.0;
          }
        }
      }
    }
  } else {
#pragma unroll
  for (int u = 0; u < kPerWave; ++u) {
    const uint32_t q = wave + 4u * (uint32_t)u;
    if (q < nblk && (q < RB || blockIdx.x == 0)) {
      const uint32_t region = q / RB, cb = q % RB, nvalid = min(32u, 2u * J - 32u * cb);
      double s;
      if constexpr (WR == 0) {
        unsigned v2[kResGroups / 2];
        alive = res_sweep<kResGroups / 2, __HIP_MEMORY_SCOPE_AGENT>(L::sums(xb, region, tag & 1u, 0), GR, 32u * cb, tag, nvalid, groups, v2, &xb->abort_word,
                                                                    p.host_error, code, ticks, lane) && alive;
        s = res_sum<kResGroups / 2>(v2, lane);
      } else if (p.xchg_gather_leaders == 0u || m == 0u) {
        // (every rank runs at least 8 workgroups -- the host checks -- so all world * 8 rows exist)
        // The ORDER of the sum is that of the context's training kernels (bit-for-bit equality of a batched validation block
        // with the entry-by-entry path): where those split level 2 into two halves (res_split: K <= 16) the total is
        // (first half: even rows + odd rows) + (second half: even + odd), a half being the WR of the training kernel --
        // 8 rows up to 2 ranks, 16 up to 4, 32 up to 8; otherwise (all even rows) + (all odd rows).
        constexpr bool kSegmented = res_blocks(L::kSegK) <= 2;
        if constexpr (WR == 32 && (kWideRow || res_seq_halves(KT) || kSegmented)) {
          // up to 64 rows: polled in two halves one after the other -- 16 row pairs in flight per lane instead of 32, which
          // these instantiations have no registers for (scratch otherwise)
          const uint32_t rows_all = p.xchg_world * (uint32_t)kResGroups;
          unsigned va[16];
          alive = res_sweep<16, __HIP_MEMORY_SCOPE_SYSTEM>(L::rank_sums(p.xchg, region, tag & 1u, 0), GR, 32u * cb, tag, nvalid, min(rows_all, 32u), va,
                                                           &xb->abort_word, p.host_error, code, ticks, lane) && alive;
          if (rows_all > 32u) {  // (uniform: 5 ... 8 ranks; the training kernels' halves are rows [0, 32) and [32, 64))
            unsigned vb[16];
            alive = res_sweep<16, __HIP_MEMORY_SCOPE_SYSTEM>(L::rank_sums(p.xchg, region, tag & 1u, 32u), GR, 32u * cb, tag, nvalid, rows_all - 32u, vb,
                                                             &xb->abort_word, p.host_error, code, ticks, lane) && alive;
            if constexpr (kSegmented) {
              s = res_segment<0, 16>(va) + res_segment<0, 16>(vb);
            } else {
              const double acc = res_rows_add<16>(vb, res_rows_add<16>(va, 0.0));
              s = pair_add<32>(acc, acc);
            }
          } else if (kSegmented && p.xchg_world > 2u) {  // 3 or 4 ranks: halves of 16 rows
            s = res_segment<0, 8>(va) + res_segment<8, 16>(va);
          } else if (kSegmented) {                       // (2 ranks run the WR = 8 instantiation; kept for completeness: halves of 8 rows)
            s = res_segment<0, 4>(va) + res_segment<4, 8>(va);
          } else {
            s = res_segment<0, 16>(va);
          }
        } else {
        unsigned v2[WR];
        alive = res_sweep<WR, __HIP_MEMORY_SCOPE_SYSTEM>(L::rank_sums(p.xchg, region, tag & 1u, 0), GR, 32u * cb, tag, nvalid,
                                                         p.xchg_world * (uint32_t)kResGroups, v2, &xb->abort_word, p.host_error, code, ticks, lane) && alive;
        if constexpr (kSegmented && (WR == 8 || WR == 16)) {
          // (the launchers pick WR = 8 up to 2 ranks and -- ts_hybhol -- 16 up to 4: the training kernels' halves are WR rows each)
          s = res_segment<0, WR / 2>(v2) + res_segment<WR / 2, WR>(v2);
        } else {
          s = res_sum<WR>(v2, lane);
        }
        }
        // three levels (TSAMD_SCHEDULE_GATHER=leaders): only the eight leaders of a rank poll the world x 8 rows; each hands
        // the total -- the same bits on every leader of every rank -- to the members of its group through its local row
        if (p.xchg_gather_leaders != 0u && lane < nvalid && !(lane & 1u)) res_post(L::sums(xb, region, tag & 1u, g) + 32u * cb + lane, tag, s, 0);
      } else {
        unsigned v1[1];
        alive = res_sweep<1, __HIP_MEMORY_SCOPE_AGENT>(L::sums(xb, region, tag & 1u, g), GR, 32u * cb, tag, nvalid, 1u, v1, &xb->abort_word, p.host_error,
                                                       code, ticks, lane) && alive;
        s = res_sum<1>(v1, lane);
      }
      sink(region, cb, nvalid, lane, s);
    }
  }
  }
  if (lane == 0 && !alive) s_alive[wave] = 0;
  __syncthreads();
  return (s_alive[0] & s_alive[1] & s_alive[2] & s_alive[3]) != 0;
}

// per-item access to the packed 2-bit column: VEC individuals per item
template <int VEC>
struct ResCodes {
  static constexpr uint32_t kItemsPerWord = 16u / VEC, kCodeBits = 2u * VEC, kMask = (1u << kCodeBits) - 1u;
  static constexpr uint32_t kMissing = VEC == 2 ? 0x5u : 0x1u;  // PLINK 01 for every individual of the item
};

// w = exp(psi(g)) up to a per-individual factor, like gamma_to_w, holding K instead of 2K temporaries across the
// maximum (z = g + 10 is formed again instead of kept): the large-K instantiations live on their registers
template <int KT>
__device__ __forceinline__ void gamma_to_w_lean(const double (&g)[KT], double (&w)[KT]) {
  double a[KT];
  double amax = -1.0e300;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    double z;
    exp_digamma_split(g[k], z, a[k]);
    amax = fmax(amax, a[k]);
  }
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    double gk = g[k];
    asm volatile("" : "+v"(gk));  // (opaque: otherwise the z of the first loop is kept alive instead)
    w[k] = (gk + 10.0) * exp_nonpos(a[k] - amax);
  }
}

// The sweep of one item: both parents' normalisers and the 2K accumulations per individual
// (update_phimom / update_phidad / update_lambda_t in the linear domain, tsamd_device.h).  BS: exp(Elogbeta) in
// registers for the whole sweep (b0 / b1: scalar registers at K <= 8, vector registers up to K = 24); otherwise
// read as (b[k][0], b[k][1]) pairs from LDS at every use.
template <int KT, int VEC, bool BS>
__device__ __forceinline__ void res_consume(const typename Lanes<VEC>::T (&wv)[KT], uint32_t code, const double (&b0)[BS ? KT : 1],
                                            const double (&b1)[BS ? KT : 1], const double2 *s_b, double (&acc0)[KT], double (&acc1)[KT]) {
  double c0[VEC], c1[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    double mom, dad;
    bool ok;
    code_weights((code >> (2 * v)) & 3u, mom, dad, ok);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      double wk;
      if constexpr (VEC == 2) wk = v ? wv[k].y : wv[k].x; else wk = wv[k];
      if constexpr (BS) {
        s0 = fma(wk, b0[k], s0);
        s1 = fma(wk, b1[k], s1);
      } else {
        const double2 b = s_b[k];
        s0 = fma(wk, b.x, s0);
        s1 = fma(wk, b.y, s1);
      }
    }
    // mom / s0 and dad / s1 through ONE reciprocal, of s0 * s1 (both are sums of positive terms of moderate size)
    const double inv = fast_rcp(s0 * s1);
    c0[v] = (mom * s1) * inv;
    c1[v] = (dad * s0) * inv;
  }
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    if constexpr (VEC == 2) {
      acc0[k] = fma(c0[1], wv[k].y, acc0[k]);
      acc1[k] = fma(c1[1], wv[k].y, acc1[k]);
      acc0[k] = fma(c0[0], wv[k].x, acc0[k]);
      acc1[k] = fma(c1[0], wv[k].x, acc1[k]);
    } else {
      acc0[k] = fma(c0[0], wv[k], acc0[k]);
      acc1[k] = fma(c1[0], wv[k], acc1[k]);
    }
  }
}

// The same for ONE individual whose genotype factors (y, 2 - y; both 0 when unobserved) are already decoded: the same
// instructions on the same values in the same order as res_consume<KT, 1, BS> -- the same bits.  ts_schedule and
// ts_holblock decode a column once per SNP / batch into nibbles (pack_nibbles) instead of once per individual and pass.
template <int KT, bool BS>
__device__ __forceinline__ void res_consume_md(const double (&wv)[KT], double mom, double dad, const double (&b0)[BS ? KT : 1],
                                               const double (&b1)[BS ? KT : 1], const double2 *s_b, double (&acc0)[KT], double (&acc1)[KT]) {
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    if constexpr (BS) {
      s0 = fma(wv[k], b0[k], s0);
      s1 = fma(wv[k], b1[k], s1);
    } else {
      const double2 b = s_b[k];
      s0 = fma(wv[k], b.x, s0);
      s1 = fma(wv[k], b.y, s1);
    }
  }
  const double inv = fast_rcp(s0 * s1);
  const double c0 = (mom * s1) * inv, c1 = (dad * s0) * inv;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    acc0[k] = fma(c0, wv[k], acc0[k]);
    acc1[k] = fma(c1, wv[k], acc1[k]);
  }
}
// PLINK 2-bit code -> nibble (y, 2 - y) in bits 0-1 / 2-3; 0 for a missing (or held-out, or unowned) genotype
__device__ __forceinline__ uint32_t code_nibble(uint32_t c) {
  const uint32_t hi = c >> 1, lo = c & 1u, miss = lo & (hi ^ 1u), y = hi * (1u + lo);  // (code_weights, tsamd_device.h)
  return y | ((2u - y - 2u * miss) << 2);
}
// nibble t of a thread's packed items (x: items 0 .. 7, y: items 8 .. 15; t is a constant after unrolling)
__device__ __forceinline__ uint32_t nibble_of(const uint2 &c, int t) {
  return ((t < 8 ? c.x : c.y) >> (4 * (t % 8))) & 15u;
}

// workgroup reduction of the 2K accumulators, fixed order: lanes (halving butterfly) -> s_red[wave][value]
template <int KT>
__device__ __forceinline__ void res_fold(const double (&acc0)[KT], const double (&acc1)[KT], double *s_red /* [4][2K] */, uint32_t tid) {
  using Fold = WaveFold<2 * KT>;
  const uint32_t lane = tid & 63u, wave = tid >> 6;
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
  if ((lane & (kRep - 1u)) == 0u && slot < 2 * KT) s_red[wave * (2 * KT) + slot] = tot;
}

// ---------------------------------------------------------------------------
// ts_resident<K>: ALL plain passes of a SNP in one launch.  The first of its passes streams the weights exactly like
// the plain pass and keeps them; every later pass runs from registers.  Same state machine as the launch-per-pass
// sequence: it starts from the first pass' State and partial rows and leaves State / partial rows for the next first
// pass (or ts_flush); the workgroups reach the complete / continue decision identically from the same totals.
// Nothing in memory is modified before the first exchange has succeeded, so a launch whose workgroups are not all
// resident (the exchange needs them all at once) gives up there with the state intact (kFailIntact).
template <int KT>
__global__ __launch_bounds__(256, 1) void ts_resident(Ctl *ctl_a, double *partials_a, double *w_a, uint32_t npad_a, uint32_t chunk_a,
                                                      uint32_t par_arg, uint32_t nrows_hint, ResXchg *xb, uint32_t serial, const DevParams p) {
  constexpr int BLOCK = 256, kWaves = BLOCK / 64, VEC = resident_vec(KT), kItems = resident_items(KT);
  // (vector registers only up to K = 11 here -- this kernel also holds the first sweep's loads in flight -- and not at K = 9, whose
  // 14 items leave no room for them either: 12 and 8 bytes of scratch at K = 9 and 12 until round 6)
  constexpr bool BS = KT <= 8 || KT == 10 || KT == 11, BSC = KT <= 8;
  using LN = Lanes<VEC>;
  using WT = typename LN::T;
  using RC = ResCodes<VEC>;
  constexpr uint32_t J = 2 * KT;
  __shared__ __attribute__((aligned(16))) double s_eb[J];
  __shared__ double s_lam[J], s_diff[J], s_tot[2 * J];
  __shared__ double s_red[kWaves * J];
  __shared__ double s_fin[BLOCK];
  __shared__ int s_alive[4];

  const uint32_t par = par_arg & 1u;
  Ctl *ctl = ctl_a;
  const State *S = &ctl->st[par ^ 1u];
  State *W = &ctl->st[par];
  const uint32_t sidx = S->idx, svalid = S->valid, sdone = S->done, sloc = S->loc, shol = S->hol;
  const uint32_t siters = S->iters, snrows = S->nrows;
  const uint32_t xseq0 = ctl->xseq;  // (workgroup 0 advances it when it leaves, after everybody's first exchange)
  const unsigned long long aborted = __hip_atomic_load(&xb->abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_sched_barrier(0);
  const uint32_t tid = threadIdx.x;
  const size_t np = npad_a;
  const uint32_t nitems = npad_a / (uint32_t)VEC;
  const uint32_t begin = blockIdx.x * chunk_a, end = min(begin + chunk_a, nitems);
  const uint32_t i0 = begin + tid;
  const uint32_t cnt = (i0 < end) ? min((end - i0 + BLOCK - 1u) / BLOCK, (uint32_t)kItems) : 0u;
  // (item t of the thread, or its last one, or -- a thread that owns none -- the shard's last item; without a select:
  // a thread that owns items has i0 < nitems)
  auto item_or_last = [&](uint32_t t) { return min(i0, nitems - 1u) + min(t, max(cnt, 1u) - 1u) * BLOCK; };
  auto load_rows = [&](uint32_t i, WT (&wv)[KT]) {
#pragma unroll
    for (int k = 0; k < KT; ++k) wv[k] = reinterpret_cast<const WT *>(w_a + (size_t)k * np)[i];
  };
  if (tid < 4) s_alive[tid] = 1;

  // the first pass' partial rows and the epilogue's inputs first, then the first item's rows
  RowSum<BLOCK> rowsum;
  rowsum.issue(partials_a + (size_t)(par ^ 1u) * kMaxGrid * J, nrows_hint, J);
  const PendingIn pin = load_pending(S, J);
  WT buf[kItems][KT];
  load_rows(item_or_last(0), buf[0]);
  __builtin_amdgcn_sched_barrier(0);
  uint32_t codes = 0u;  // the items' 2-bit codes of this SNP's column, packed (an item no thread owns: missing)
  {
    const uint32_t *col = reinterpret_cast<const uint32_t *>(p.bed + (size_t)sloc * p.colstride);
    uint32_t word[kItems];
#pragma unroll
    for (int t = 0; t < kItems; ++t) word[t] = col[item_or_last((uint32_t)t) / RC::kItemsPerWord];
#pragma unroll
    for (int t = 0; t < kItems; ++t) {
      const uint32_t i = item_or_last((uint32_t)t);
      const uint32_t c = (uint32_t)t < cnt ? (word[t] >> (RC::kCodeBits * (i % RC::kItemsPerWord))) & RC::kMask : RC::kMissing;
      codes |= c << (RC::kCodeBits * (uint32_t)t);
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  if (aborted != 0ull) return;  // an earlier launch of this context gave up: nothing runs until the host has dealt with it
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
  double b0[BS ? KT : 1], b1[BS ? KT : 1], acc0[KT], acc1[KT];
  // start of a pass: the values the previous epilogue left in LDS become this pass' inputs
  auto begin_pass = [&]() {
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
  auto consume = [&](int t, const WT (&wv)[KT]) {
    uint32_t zo = 0u;  // (opaque zero: the LDS reads of exp(Elogbeta) are repeated per item instead of held in 4K registers)
    if constexpr (!BS) asm volatile("" : "+v"(zo));
    res_consume<KT, VEC, BS>(wv, (codes >> (RC::kCodeBits * (uint32_t)t)) & RC::kMask, b0, b1, reinterpret_cast<const double2 *>(s_eb) + zo, acc0,
                             acc1);
  };
  // end of a pass: workgroup reduction; then either (the cap) hand rows and state to the next launch,
  // or exchange the rows inside the launch and run the epilogue.  Returns true when the kernel is over.
  uint32_t xcount = 0u;  // exchanges of this launch
  auto finish_pass = [&]() -> bool {
    res_fold<KT>(acc0, acc1, s_red, tid);
    __syncthreads();
    double row = 0.0;
    if (tid < J) {
      row = s_red[tid];
#pragma unroll
      for (int wv = 1; wv < kWaves; ++wv) row += s_red[wv * J + tid];
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
    xcount += 1u;
    const uint32_t tag = xseq0 + xcount;
    // (nothing has been written to memory before the launch's first exchange: a failure there leaves the state intact)
    // (one level only up to 16 workgroups here, and only at K <= 8: the sweep's registers do not fit beside this kernel's)
    if (!res_exchange<KT, 0, (KT <= 8 ? 16 : 0)>(xb, p, tag, 1u, row, g, m, gridDim.x, s_tot, s_alive, tid, fail_code(tag, xcount == 1u, par, serial),
                             xcount == 1u ? (unsigned long long)p.probe_ticks : kResWaitTicks))
      return true;  // (the abort word is set: tsamd_synchronize deals with it)
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
  if constexpr (kItems > 1) load_rows(item_or_last(1u), buf[1]);
#pragma unroll
  for (int t = 0; t < kItems; ++t) {
    if (t + 2 < kItems) load_rows(item_or_last((uint32_t)t + 2u), buf[t + 2]);
    __builtin_amdgcn_sched_barrier(0);
    consume(t, buf[t]);
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
      consume(t, buf[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (finish_pass()) return;
  }
}

// ---------------------------------------------------------------------------
// ts_schedule<K, PARTIAL, WR>: a WHOLE schedule in one launch (one GPU, or -- WR > 0 -- one launch per rank of a
// sharded run).  The weights stay in registers from the first SNP to the last: the gamma step of a SNP reads and
// writes gamma (and c_n) only -- about half of it from LDS -- and overwrites the registers with the new weights, every
// pass runs from registers and ends with the in-launch exchange.  Per SNP the memory traffic drops from (I + 3) 8NK
// to 8NK .. 16NK; w and the LDS-resident gamma are written back once, at the end of the launch.
//   * The launch begins with an EMPTY exchange before it touches anything: if not all its workgroups are resident it
//     gives up there, the state intact (kFailIntact), and the host replays the schedule one launch per pass.
//   * The row of a SNP's LAST pass under the pass cap (known to be the last before it starts) is not exchanged on the
//     spot: the next SNP's gamma step needs only the exp(Elogbeta) that pass used.  It travels as region B of the next
//     SNP's first exchange, after which workgroup 0 alone runs that SNP's final epilogue (on its second wave, beside
//     the first wave's epilogue of the new pass) and publishes lambda -- one exchange latency less per SNP.  Not when
//     one of the next two SNPs revisits the location (they need the final lambda at once) or the launch ends.
// Same semantics as the launch-per-pass state machine: starts from the State the previous call left (its last SNP
// complete, its gamma step possibly pending) and leaves such a State.  PARTIAL: skip the item bodies no thread of
// the workgroup owns (small shards).
// How many of a thread's items keep their gamma in LDS (per item: K rows x 8 VEC bytes + c_n for 256 threads), and
// which streamed item follows item t (items = none)
constexpr int sched_lds_items(int k, int items, int vec) {
#ifdef TSAMD_SCHED_LDS_ITEMS  // (experiments, tools/variant.sh)
  return TSAMD_SCHED_LDS_ITEMS < items ? TSAMD_SCHED_LDS_ITEMS : items;
#else
  const int small = 1536 + 26 * 2 * k * 8;  // the K x 2 arrays below
  const int per_item = (k * 8 + 4) * vec * 256, n = (160 * 1024 - small) / per_item;
  return n < items ? n : items;
#endif
}
constexpr int sched_next_streamed(int t, int lds, int items) {
  for (int u = t + 1; u < items; ++u)
    if (((u + 1) * lds) / items == (u * lds) / items) return u;
  return items;
}

template <int KT, bool PARTIAL, int WR>
__global__ __launch_bounds__(256, 1) void ts_schedule(Ctl *ctl_a, double *w_a, uint32_t npad_a, uint32_t chunk_a, uint32_t par_arg,
                                                      const uint32_t *sched, uint32_t n_sched, ResXchg *xb, uint32_t serial, const DevParams p) {
  constexpr int BLOCK = 256, kWaves = BLOCK / 64, VEC = resident_vec(KT), kItems = sched_items(KT, WR);
  constexpr bool BS = KT <= 24, BSC = KT <= 8;  // exp(Elogbeta) of a pass in registers / in scalar registers
  using LN = Lanes<VEC>;
  using WT = typename LN::T;
  using CT = typename LN::C;
  using RC = ResCodes<VEC>;
  constexpr uint32_t J = 2 * KT;
  static_assert(VEC == 1 && kItems <= 16, "one individual per item; the items' genotype nibbles are packed into two registers");
  __shared__ __attribute__((aligned(16))) double s_eb[J];
  __shared__ __attribute__((aligned(16))) double s_sb[J];
  __shared__ double s_lam[J], s_diff[J], s_tot[res_tot_doubles<KT, WR>()], s_plam[J], s_peb[J];
  __shared__ double s_drow[J], s_dlam[J], s_dolam[J], s_doeb[J], s_ddiff[J];  // the deferred last pass of the previous SNP
  // kRepl: every WAVE runs the K x 2 epilogue of a pass for itself on lanes < 2K (same totals, same code, same bits) and
  // keeps lambda / exp(Elogbeta) of the pending pass in those lanes' registers: no workgroup barrier between the exchange
  // and the next sweep, nobody waits for wave 0 (-7 % per update below ~250K individuals, where an update IS its
  // exchanges and epilogues).  exp(Elogbeta) and the |dlambda| terms go through a per-wave LDS row only to be broadcast to
  // the wave's other lanes (rounds 3-4; round 5: exp(Elogbeta) goes from the lanes' registers to scalar registers with
  // v_readlane, no LDS row).  K <= 8 only -- every instantiation since round 5; above, the shared form: threads < 2K, shared
  // arrays, a barrier.
#if defined(TSAMD_SHARED_EPILOGUE)  // (experiments)
  constexpr bool kRepl = false;
#else
  // (K > 8: measured neutral to 1.5 % slower -- wider rows, more lanes in the epilogue.  Round 5: with exp(Elogbeta) taken from the
  // lanes' registers -- TSAMD_REPL_READLANE -- the per-wave form wins on the full-size instantiation as well: N = 1M, K = 8
  // 72.7-73.0 against 73.6-74.0 us per update; until then it ran only below full size)
  constexpr bool kRepl = KT <= 8;
#endif
  __shared__ __attribute__((aligned(16))) double s_ebw[kWaves][J];
  __shared__ double s_diffw[kWaves][J];
  __shared__ double s_red[kWaves * J];
  __shared__ int s_alive[4];
  // gamma (and c_n) of kLds of a thread's items stay in LDS for the whole launch, spread evenly over the items; the
  // gamma step streams the others from memory, one streamed item ahead.  Memory sees them again when the launch ends.
  constexpr int kLds = sched_lds_items(KT, kItems, VEC);
  __shared__ WT s_gam[kLds > 0 ? kLds : 1][KT][BLOCK];
  __shared__ CT s_cn[kLds > 0 ? kLds : 1][BLOCK];
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
  const uint32_t nitems = npad_a / (uint32_t)VEC;
  const uint32_t begin = blockIdx.x * chunk_a, end = min(begin + chunk_a, nitems);
  uint32_t i0 = begin + tid;
  uint32_t cnt = (i0 < end) ? min((end - i0 + BLOCK - 1u) / BLOCK, (uint32_t)kItems) : 0u;
  // Everything below sits in one loop over the schedule with the sweeps fully unrolled: left alone, the
  // compiler hoists every address that depends only on (thread, item, row) out of that loop -- a few
  // hundred values, spilled -- so the three values they derive from are made opaque per use.
  auto fresh = [&]() { asm volatile("" : "+v"(tid), "+v"(i0), "+v"(cnt)); };
  // (item t of the thread, or its last one, or -- a thread that owns none -- the shard's last item; without a select:
  // a thread that owns items has i0 < nitems)
  auto item_or_last = [&](uint32_t t) { return min(i0, nitems - 1u) + min(t, max(cnt, 1u) - 1u) * BLOCK; };
  // items any thread of this workgroup owns (uniform).  PARTIAL (the host picks it when a workgroup's chunk leaves
  // whole items unused: shards well below the capacity): the item bodies nobody needs are skipped; the branches
  // cost the full-size kernel 5 %, so it runs without them -- an unused item is then processed as "missing".
  const uint32_t cnt_wg = !PARTIAL ? (uint32_t)kItems : begin < end ? min((end - begin + BLOCK - 1u) / BLOCK, (uint32_t)kItems) : 0u;
  const uint32_t g = blockIdx.x % (uint32_t)kResGroups, m = blockIdx.x / (uint32_t)kResGroups;

  if (__hip_atomic_load(&xb->abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) return;  // (see ts_resident)
  if (n_sched == 0u) {
    if (blockIdx.x == 0) carry_state(S, W, J);
    return;
  }
  if (tid < 4) s_alive[tid] = 1;
  __syncthreads();
  uint32_t xcount = 1u;  // exchanges of this launch
  // the entry exchange: empty rows, nothing modified yet.  All workgroups resident?
  if (!res_exchange<KT, WR>(xb, p, xseq0 + 1u, 1u, 0.0, g, m, gridDim.x, s_tot, s_alive, tid, fail_code(xseq0 + 1u, true, par, serial),
                            WR == 0 ? (unsigned long long)p.probe_ticks : kResWaitTicks))
    return;
  if constexpr (WR > 0) {
    // Sharded: the entry exchange's verdict is not collective -- a rank whose last workgroup gave up a moment before the
    // others' rows arrived fails while its peers, which had its leaders' sums already, pass.  A second empty exchange
    // commits it: a rank that failed the first never posts here, so nobody completes it and EVERY rank gives up with its
    // state intact (and replays, csrc/tsamd.hip).  Bounded by a third of the entry's wait: every rank that passed the
    // first exchange posts within microseconds, and the ranks' replays must start within one peer-to-peer wait of each other.
    xcount += 1u;
    if (p.xchg_test_delay != 0u) {  // test hook (TSAMD_TEST_XCHG_DELAY_US, contexts created with TSAMD_FLAG_TEST_HOOKS only): this rank posts its commit late
      const unsigned long long t0 = wall_clock64();
      while (wall_clock64() - t0 < (unsigned long long)p.xchg_test_delay) __builtin_amdgcn_s_sleep(8);
    }
    if (!res_exchange<KT, WR>(xb, p, xseq0 + xcount, 1u, 0.0, g, m, gridDim.x, s_tot, s_alive, tid, fail_code(xseq0 + xcount, true, par, serial),
                              kResWaitTicks / 3ull))
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
      s_cn[lds_slot(t)][tid] = reinterpret_cast<const CT *>(p.cnt)[i];
    }
  auto get_item = [&](int t, WT (&wv)[KT]) {
#pragma unroll
    for (int k = 0; k < KT; ++k) wv[k] = buf[t][k];
  };
  auto put_item = [&](int t, const WT (&wv)[KT]) {
#pragma unroll
    for (int k = 0; k < KT; ++k) buf[t][k] = wv[k];
  };
  // gamma and c_n of an LDS item
  auto get_lgamma = [&](int t, WT (&gv)[KT], CT &cv) {
#pragma unroll
    for (int k = 0; k < KT; ++k) gv[k] = s_gam[lds_slot(t)][k][tid];
    cv = s_cn[lds_slot(t)][tid];
  };
  auto put_lgamma = [&](int t, const WT (&gv)[KT], const CT &cv) {
#pragma unroll
    for (int k = 0; k < KT; ++k) s_gam[lds_slot(t)][k][tid] = gv[k];
    s_cn[lds_slot(t)][tid] = cv;
  };
  // the genotypes of a column for this thread's items, decoded once per SNP into nibbles (y, 2 - y; 0 for a missing or
  // held-out genotype and for an item the thread does not own) and packed into two registers: a sweep then spends two
  // bit-field extracts and two conversions per individual where code_weights spends ten instructions, ten times per
  // SNP on the same codes.  In two steps, so that the words of the NEXT SNP's column can be requested a SNP ahead and
  // packed when they are needed.
  auto load_words = [&](uint32_t loc_, auto &word) {
    const uint32_t *col = reinterpret_cast<const uint32_t *>(p.bed + (size_t)loc_ * p.colstride);
#pragma unroll
    for (int t = 0; t < kItems; ++t) word[t] = col[item_or_last((uint32_t)t) / RC::kItemsPerWord];
  };
  auto pack_codes = [&](const auto &word) -> uint2 {
    uint32_t out[2] = {0u, 0u};
#pragma unroll
    for (int t = 0; t < kItems; ++t) {
      const uint32_t i = item_or_last((uint32_t)t);
      const uint32_t c = (uint32_t)t < cnt ? (word[t] >> (RC::kCodeBits * (i % RC::kItemsPerWord))) & RC::kMask : RC::kMissing;
      out[t / 8] |= code_nibble(c) << (4u * (uint32_t)(t % 8));
    }
    return make_uint2(out[0], out[1]);
  };
  auto load_codes = [&](uint32_t loc_) -> uint2 {
    uint32_t word[kItems];
    load_words(loc_, word);
    return pack_codes(word);
  };
  // the previous call's last SNP: its gamma step may be pending (column bits, exp(Elogbeta) of its last
  // pass), and its final values serve a first SNP at the same location
  uint2 pcodes = svalid ? load_codes(sloc) : make_uint2(0u, 0u);
  // (per-wave form: lane j < 2K of every wave also keeps the exp(Elogbeta) the last executed pass used in a register -- eb_ran --,
  // from which the gamma step takes it with v_readlane)
  const double sb_state = S->eb[(tid & 63u) < J ? (tid & 63u) : 0u];
  if (tid < J) {
    s_sb[tid] = S->eb[tid];
    s_plam[tid] = svalid ? p.lam[(size_t)sloc * J + tid] : 0.0;
    s_peb[tid] = svalid ? p.eb[(size_t)sloc * J + tid] : 0.0;
  }
  bool do_gamma = svalid != 0u && shol == 0u;
  bool prev_valid = svalid != 0u;
  uint32_t prev_loc = sloc, prev_hol = shol, prev_iters = siters;
  bool w_dirty = false;
  bool deferred = false;  // the previous SNP's last row waits in s_drow for this SNP's first exchange
  uint32_t dloc = 0u, diters = 0u;
  bool pub_pending = false;  // workgroup 0: lambda of a SNP published since its first wave last waited for its stores
  // The SNP counters (count_snp in the launch-per-pass kernels): the histogram bin is bumped by a fire-and-forget
  // atomic, the totals run in registers of workgroup 0's thread 0 and go to memory -- and to the pinned host mirror --
  // when the launch ends: no dependent memory round trip on the publishing workgroup's critical path per SNP.
  unsigned long long tp_run = ctl->total_passes;
  uint32_t last_it = ctl->last_iters;
  auto count_snp_deferred = [&](uint32_t its) {
    const uint32_t bin = min(its, (uint32_t)TSAMD_PASS_HIST_BINS - 1u);
    __hip_atomic_fetch_add(&ctl->pass_hist[bin], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (result unused: no return, no wait)
    tp_run += (unsigned long long)its;
    last_it = its;
  };
#ifdef TSAMD_SCHED_RAMP
  unsigned long long ramp_mark = wall_clock64();
  uint32_t ramp_idx = 0u;
#endif
#ifdef TSAMD_SCHED_TIME  // diagnostic build (tools/variant.sh): where a SNP's time goes, 10 ns ticks, workgroup 0
  unsigned long long tk_gamma = 0, tk_first = 0, tk_rest = 0, tk_xchg = 0, tk_head = 0, tk_tail = 0, tk_mark = wall_clock64();
  unsigned long long tk_fold = 0, tk_epi = 0, tk_sweep = 0;
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
  double b0[BS ? KT : 1], b1[BS ? KT : 1], acc0[KT], acc1[KT];
  bool complete = false;
  // every later SNP's column is requested while its predecessor runs -- except where the instantiation has no room for the
  // words in flight (K = 16 on one GPU: 128 resident doubles per thread, 36 bytes of scratch otherwise; K = 14: 126 doubles,
  // 12 ... 44 bytes; K = 24 on two ranks: 12 bytes -- profiles/r06_kernel_resources.txt): those load them when the SNP starts
  constexpr bool kColAhead = !(KT == 16 || KT == 14 || (KT == 24 && WR == 8)) && KT != TSAMD_NOCOL_K;
  uint2 codes = kColAhead ? load_codes(sched[0] & 0x7fffffffu) : make_uint2(0u, 0u);
  uint32_t nword[kColAhead ? kItems : 1];
  // (default form: lam_old / eb_used live in lanes < 2K of EVERY wave and are advanced by the wave's own epilogue: lambda
  // before the pending pass' epilogue, exp(Elogbeta) of the pass that runs; eb_ran = what the last executed pass used)
  double eb_ran = sb_state;
  auto begin_pass = [&]() {
    fresh();
    iters += 1u;
    const double *bsrc = kRepl ? s_ebw[tid >> 6] : s_eb;
    if constexpr (kRepl) {
      eb_ran = eb_used;  // (the epilogue advances eb_used; the next SNP's gamma step needs what the LAST pass used)
    } else {
      lam_old = s_lam[tid < J ? tid : 0u];
      eb_used = s_eb[tid < J ? tid : 0u];
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      if constexpr (kRepl && BSC && TSAMD_REPL_READLANE) {
        // per-wave form: lane j < 2K of THIS wave holds exp(Elogbeta_j) of the pass in eb_used -- taken straight from the
        // lanes' registers (v_readlane) instead of through the wave's LDS row (a store, a wait, eight 16-byte reads)
        b0[k] = lane_f64(eb_used, 2 * k);
        b1[k] = lane_f64(eb_used, 2 * k + 1);
      } else if constexpr (BSC) {
        b0[k] = uniform_f64(bsrc[2 * k]);
        b1[k] = uniform_f64(bsrc[2 * k + 1]);
      } else if constexpr (BS) {
        b0[k] = bsrc[2 * k];
        b1[k] = bsrc[2 * k + 1];
      }
      acc0[k] = acc1[k] = 0.0;
    }
  };
  auto sweep = [&]() {
#ifdef TSAMD_SCHED_TIME
    const unsigned long long ts0 = wall_clock64();
#endif
#pragma unroll
    for (int t = 0; t < kItems; ++t) {
      if (PARTIAL && (uint32_t)t >= cnt_wg) continue;
      fresh();
      WT wcur[KT];
      get_item(t, wcur);
      uint32_t zo = 0u;  // (opaque zero: the LDS reads of exp(Elogbeta) are repeated per item instead of held in 4K registers)
      if constexpr (!BS) asm volatile("" : "+v"(zo));
      const double2 *bl = reinterpret_cast<const double2 *>(kRepl ? s_ebw[tid >> 6] : s_eb) + zo;
      {
        const uint32_t nib = nibble_of(codes, t);
        res_consume_md<KT, BS>(wcur, (double)(nib & 3u), (double)(nib >> 2), b0, b1, bl, acc0, acc1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef TSAMD_SCHED_TIME
    tk_sweep += wall_clock64() - ts0;
#endif
  };
  // end of a pass: workgroup reduction, then the in-launch exchange and the epilogue -- or, defer == true, the row is
  // parked for the next SNP's first exchange.  false: the exchange gave up.
  auto finish_pass = [&](bool defer) -> bool {
    fresh();
#ifdef TSAMD_SCHED_TIME
    const unsigned long long tf0 = wall_clock64();
#endif
    // workgroup 0 publishes a deferred SNP's final lambda / exp(Elogbeta) right after a first exchange (kRepl: its second
    // wave) and must have them out before it joins the next one: every wave waits for its own stores here, a whole sweep
    // later (it costs nothing), ahead of the barrier that precedes the post
    if (kRepl && pub_pending) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      pub_pending = false;
    }
    res_fold<KT>(acc0, acc1, s_red, tid);
    __syncthreads();
#ifdef TSAMD_SCHED_TIME
    const unsigned long long tx0 = wall_clock64();
    tk_fold += tx0 - tf0;
#endif
    double mine = 0.0;
    if (tid < J) {
      mine = s_red[tid];
#pragma unroll
      for (int wv = 1; wv < kWaves; ++wv) mine += s_red[wv * J + tid];
    }
    if (defer) {  // the SNP's last pass under the cap: its row and its epilogue's inputs wait for the next exchange
      if (tid < J) s_drow[tid] = mine;
      if (tid < J) s_dlam[tid] = lam_old;
      complete = true;
      return true;  // (the caller's end-of-SNP barrier orders these stores)
    }
    const uint32_t width = deferred ? 2u : 1u;
    if (deferred && tid >= 64u && tid < 64u + J) mine = s_drow[tid - 64u];
    // workgroup 0 publishes a SNP's final lambda / exp(Elogbeta) with plain agent-scope stores (threads < J, its first
    // wave) and must have them out before it joins the next exchange: whoever completes that exchange may read them.
    // A deferred SNP is published right after a first exchange; the wait stands here, a whole sweep later, where it
    // costs nothing.
    if (!kRepl && pub_pending) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      pub_pending = false;
    }
    xcount += 1u;
    const uint32_t tag = xseq0 + xcount;
    if (!res_exchange<KT, WR>(xb, p, tag, width, mine, g, m, gridDim.x, s_tot, s_alive, tid, fail_code(tag, false, par, serial), kResWaitTicks))
      return false;
#ifdef TSAMD_SCHED_TIME
    const unsigned long long te0 = wall_clock64();
    tk_xchg += te0 - tx0;
#endif
    if constexpr (kRepl) {
      const uint32_t lane = tid & 63u, wave = tid >> 6;
      if (lane < J) {
        double nw, ebn, df;
        epilogue_values_reg(p, lane, res_total<KT, WR>(s_tot, lane), eb_used, lam_old, nw, ebn, df);
        lam_old = nw;
        eb_used = ebn;
        if constexpr (!(BSC && TSAMD_REPL_READLANE)) s_ebw[wave][lane] = ebn;
        s_diffw[wave][lane] = df;
      }
      if (deferred) {
        if (blockIdx.x == 0) {  // the previous SNP's final epilogue and its publication: the second wave, from registers
          if (wave == 1u && lane < J) {
            double nw, ebn, df;
            epilogue_values_reg(p, lane, res_total<KT, WR>(s_tot, J + lane), s_sb[lane], s_dlam[lane], nw, ebn, df);
            __hip_atomic_store(&p.lam[(size_t)dloc * J + lane], nw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&p.eb[(size_t)dloc * J + lane], ebn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (tid == 0) count_snp_deferred(diters);
          pub_pending = true;  // (out before this workgroup joins the next exchange: the wait at the top of finish_pass)
        }
        deferred = false;
      }
      complete = epilogue_complete(p, iters, J, s_diffw[wave]);
    } else {
    if (tid < J) epilogue_values(p, res_total<KT, WR>(s_tot, tid), eb_used, lam_old, s_lam, s_eb, s_diff);
    if (deferred && blockIdx.x == 0 && tid >= 64u && tid < 64u + J)  // the previous SNP's final epilogue, beside the new pass' one
      epilogue_values_at(p, tid - 64u, res_total<KT, WR>(s_tot, J + tid - 64u), s_sb[tid - 64u], s_dlam[tid - 64u], s_dolam, s_doeb, s_ddiff);
    __syncthreads();
    if (deferred) {
      if (blockIdx.x == 0) {  // publish the previous SNP (before this workgroup joins the next exchange)
        if (tid < J) {
          __hip_atomic_store(&p.lam[(size_t)dloc * J + tid], s_dolam[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&p.eb[(size_t)dloc * J + tid], s_doeb[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) count_snp_deferred(diters);
        pub_pending = true;  // (out before this workgroup joins the next exchange: the wait above, one sweep from now)
      }
      deferred = false;
    }
    if constexpr (KT > 8) complete = epilogue_complete_wave<J>(p, iters, s_diff, tid & 63u); else complete = epilogue_complete(p, iters, J, s_diff);
    }
#ifdef TSAMD_SCHED_TIME
    tk_epi += wall_clock64() - te0;
#endif
    return true;
  };

  // The next SNP's entry and its location's lambda / exp(Elogbeta) are requested one SNP ahead (after the
  // current SNP's first exchange: workgroup 0 has then published every earlier SNP of this launch with
  // agent-scope stores -- a deferred one during that very exchange, which is why a SNP is only deferred when neither
  // of the next two revisits its location); a SNP at the location of its predecessor takes that one's final values from LDS.
  uint32_t loc = 0, hol = 0;
  uint32_t ent_n = sched[0], ent_nn = sched[min(1u, n_sched - 1u)];
  double nlam = 0.0, neb = 0.0;
  if (tid < J) {
    nlam = p.lam[(size_t)(ent_n & 0x7fffffffu) * J + tid];
    neb = p.eb[(size_t)(ent_n & 0x7fffffffu) * J + tid];
  }
  for (uint32_t idx = 0; idx < n_sched; ++idx) {
    const uint32_t ent = ent_n;
    loc = ent & 0x7fffffffu;
    hol = ent >> 31;
    ent_n = ent_nn;
    ent_nn = sched[min(idx + 2u, n_sched - 1u)];
    if (tid < J) {
      const bool local = prev_valid && loc == prev_loc;
      s_lam[tid] = local ? s_plam[tid] : nlam;
      s_eb[tid] = local ? s_peb[tid] : neb;
    }
    fresh();
    if constexpr (!kColAhead) codes = load_codes(loc);
    __syncthreads();
    if constexpr (kRepl) {
      const uint32_t lane = tid & 63u, wave = tid >> 6;
      lam_old = s_lam[lane < J ? lane : 0u];
      eb_used = s_eb[lane < J ? lane : 0u];
      if constexpr (!(BSC && TSAMD_REPL_READLANE))
        if (lane < J) s_ebw[wave][lane] = eb_used;
    }
    iters = 0u;
    TSAMD_TK(tk_head);
    // ---- the previous SNP's gamma step (phi from the resident weights and the exp(Elogbeta) of that
    // SNP's last pass, read from LDS at each use).  Straight-line per item: an item past the end of the
    // thread's range is processed with "missing" codes and only its stores are guarded; an unobserved
    // genotype takes the same instructions with a step size of exactly 0.
    if (do_gamma) {
      WT gs[KT];  // the streamed item in flight (requested one streamed item ahead)
      CT cs;
      auto load_gamma = [&](uint32_t i, WT (&gq)[KT], CT &cq) {
#pragma unroll
        for (int k = 0; k < KT; ++k) gq[k] = reinterpret_cast<const WT *>(p.gam + (size_t)k * np)[i];
        cq = reinterpret_cast<const CT *>(p.cnt)[i];
      };
      // one individual: update_gamma + update_rho_indiv (src/snpsamplinge.cc:688-719) with nodekappa = 0.5 (the host
      // selects this kernel only then), then the new weights.  An unobserved genotype takes the same instructions
      // with a step size of exactly 0: gamma keeps its bits (its update term is finite), the weights are recomputed
      // from the unchanged gamma, c_n does not count -- no select per value.
#if defined(TSAMD_LITERAL_STEP)  // (experiments)
      constexpr bool kLeanStep = false;
#elif defined(TSAMD_LEAN_STEP)
      constexpr bool kLeanStep = true;
#else
      constexpr bool kLeanStep = PARTIAL || KT > 8;
#endif
      constexpr bool kSbScalar = kRepl && BSC && TSAMD_GAMMA_SB_READLANE != 0;
      double sbs[kSbScalar ? J : 1];
      if constexpr (kSbScalar) {
#pragma unroll
        for (int j = 0; j < (int)J; ++j) sbs[j] = lane_f64(eb_ran, j);
      }
      auto gamma_one = [&](double (&gx)[KT], double (&wx)[KT], uint32_t nib, uint32_t &cn) {
        const double mom = (double)(nib & 3u), dad = (double)((nib >> 2) & 3u);
        const bool ok = (nib & 15u) != 0u;  // (an observed genotype has y + (2 - y) = 2)
        double s0 = 0.0, s1 = 0.0;
        uint32_t zo = 0u;  // (opaque zero: exp(Elogbeta) is re-read from LDS where it is used, not held in 4K registers)
        asm volatile("" : "+v"(zo));
        const double *sbv = s_sb + zo;
        // (per-wave form, K <= 8: the pairs sit in scalar registers for the whole step -- sbs, below -- instead of being re-read
        // from LDS per item: every item's first instructions waited for those reads)
        auto sb = [&](int j) -> double {
          if constexpr (kSbScalar) return sbs[j]; else return sbv[j];
        };
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          s0 = fma(wx[k], sb(2 * k), s0);
          s1 = fma(wx[k], sb(2 * k + 1), s1);
        }
        const double rho = ok ? fast_rsqrt(p.nodetau0 + (double)cn) : 0.0;
        if constexpr (kLeanStep) {
          // gamma += rho (alpha + scale (y phi_mom + (2 - y) phi_dad) - gamma) as (1 - rho) gamma + rho alpha +
          // w_k (c0 sb0_k + c1 sb1_k) with rho * scale folded into c0 / c1 and ONE reciprocal (of s0 * s1) for both
          // parents: four instructions per population.  rho = 0 leaves gamma's bits alone (1 * gamma + 0 + w * 0).
          const double inv = fast_rcp(s0 * s1) * (rho * p.gamma_scale);
          const double c0 = (mom * s1) * inv, c1 = (dad * s0) * inv, keep = 1.0 - rho, ra = rho * p.alpha;
#pragma unroll
          for (int k = 0; k < KT; ++k) gx[k] = fma(wx[k], fma(c0, sb(2 * k), c1 * sb(2 * k + 1)), fma(keep, gx[k], ra));
        } else {  // (the literal form, seven instructions per population: the full-size K <= 8 instantiation spills with the other)
          const double c0 = mom * fast_rcp(s0), c1 = dad * fast_rcp(s1);
#pragma unroll
          for (int k = 0; k < KT; ++k) {
            const double e = c0 * (wx[k] * sb(2 * k)) + c1 * (wx[k] * sb(2 * k + 1));
            gx[k] += rho * (p.alpha + p.gamma_scale * e - gx[k]);
          }
        }
        if constexpr (KT <= 8) gamma_to_w<KT>(gx, wx); else gamma_to_w_lean<KT>(gx, wx);
        cn = ok ? cn + 1u : cn;
      };
      constexpr int kFirstStreamed = sched_next_streamed(-1, kLds, kItems);
      if (kFirstStreamed < kItems) load_gamma(item_or_last((uint32_t)kFirstStreamed), gs, cs);
#pragma unroll
      for (int t = 0; t < kItems; ++t) {
        if (PARTIAL && (uint32_t)t >= cnt_wg) continue;
        fresh();
        const uint32_t i = item_or_last((uint32_t)t);
        const bool mine = (uint32_t)t < cnt;
        WT gv[KT];
        CT cv;
        if (is_lds(t)) {
          get_lgamma(t, gv, cv);
        } else {
#pragma unroll
          for (int k = 0; k < KT; ++k) gv[k] = gs[k];
          cv = cs;
          constexpr int kNone = kItems;
          const int nxt = sched_next_streamed(t, kLds, kItems);
          if (nxt < kNone) load_gamma(item_or_last((uint32_t)nxt), gs, cs);
        }
        __builtin_amdgcn_sched_barrier(0);
        WT wcur[KT];
        get_item(t, wcur);
        uint32_t pcode = nibble_of(pcodes, t);
        if constexpr (VEC == 2) {
          // the item's two individuals, one after the other through ONE copy of the code (a rolled loop that
          // works on the .x halves and swaps the halves after each turn: eight items times two individuals of
          // straight-line digamma / exp code would not fit the instruction cache)
#pragma unroll 1
          for (int v = 0; v < 2; ++v) {
            double gx[KT], wx[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) {
              gx[k] = gv[k].x;
              wx[k] = wcur[k].x;
            }
            uint32_t cn = cv.x;
            gamma_one(gx, wx, pcode, cn);
            pcode >>= 2;
            cv.x = cv.y;
            cv.y = cn;
#pragma unroll
            for (int k = 0; k < KT; ++k) {
              gv[k].x = gv[k].y;
              gv[k].y = gx[k];
              wcur[k].x = wcur[k].y;
              wcur[k].y = wx[k];
            }
          }
        } else {
          gamma_one(gv, wcur, pcode, cv);
        }
        if (is_lds(t)) {
          put_lgamma(t, gv, cv);
        } else if (mine) {
#pragma unroll
          for (int k = 0; k < KT; ++k) reinterpret_cast<WT *>(p.gam + (size_t)k * np)[i] = gv[k];
          reinterpret_cast<CT *>(p.cnt)[i] = cv;
        }
        put_item(t, wcur);
        __builtin_amdgcn_sched_barrier(0);
      }
      w_dirty = true;
    }
    TSAMD_TK(tk_gamma);
    // ---- first pass of the new SNP, from the resident weights like every later one --------------------
    begin_pass();
    sweep();
    if (!finish_pass(false)) return;
    TSAMD_TK(tk_first);
    fresh();
    if (tid < J) {
      nlam = __hip_atomic_load(&p.lam[(size_t)(ent_n & 0x7fffffffu) * J + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      neb = __hip_atomic_load(&p.eb[(size_t)(ent_n & 0x7fffffffu) * J + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (kColAhead) load_words(ent_n & 0x7fffffffu, nword);  // (the next SNP's column: packed at the end of this SNP)
    while (!complete) {
      begin_pass();
      sweep();
#ifdef TSAMD_NO_DEFER
      const bool defer = false;
#else
      // under the cap this pass is known to be the SNP's last before it starts
      const bool defer = iters >= p.max_inner && idx + 1u < n_sched && (ent_n & 0x7fffffffu) != loc &&
                         (idx + 2u >= n_sched || (ent_nn & 0x7fffffffu) != loc);
#endif
      if (!finish_pass(defer)) return;
      if (defer) {
        deferred = true;
        dloc = loc;
        diters = iters;
      }
    }
    TSAMD_TK(tk_rest);
    // ---- the SNP is complete: s_lam / s_eb hold its final values (unless deferred: nobody needs them before
    // workgroup 0 has published them), eb_used the exp(Elogbeta) its last pass used.  Workgroup 0 publishes;
    // everybody keeps what the next SNP's gamma step needs.
    // (kRepl: threads < 2K are lanes < 2K of the first wave, whose registers hold the final values)
    const double fin_lam = kRepl ? lam_old : s_lam[tid < J ? tid : 0u], fin_eb = kRepl ? eb_used : s_eb[tid < J ? tid : 0u];
    if (!deferred && blockIdx.x == 0) {
      if (tid < J) {
        __hip_atomic_store(&p.lam[(size_t)loc * J + tid], fin_lam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&p.eb[(size_t)loc * J + tid], fin_eb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (tid == 0) count_snp_deferred(iters);
      // published before this workgroup joins the next exchange (not left to the wait in finish_pass: that one would
      // stand behind the next gamma step's stores)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid < J) s_sb[tid] = kRepl ? eb_ran : eb_used;
    if (tid < J) {
      s_plam[tid] = fin_lam;  // (deferred: not the final values, and never read -- the next SNP is elsewhere)
      s_peb[tid] = fin_eb;
    }
    pcodes = codes;
    fresh();
    if constexpr (kColAhead) codes = pack_codes(nword);
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
  // (the launch's last SNP is never deferred, so s_plam / s_peb / s_sb describe it completely)
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
          WT gv[KT];
          CT cv;
          get_lgamma(t, gv, cv);
#pragma unroll
          for (int k = 0; k < KT; ++k) reinterpret_cast<WT *>(p.gam + (size_t)k * np)[i] = gv[k];
          reinterpret_cast<CT *>(p.cnt)[i] = cv;
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
      ctl->total_passes = tp_run;
      ctl->last_iters = last_it;
      if (p.host_error) {
        __hip_atomic_store(p.host_error + 1, (unsigned long long)last_it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(p.host_error + 2, tp_run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the histogram atomics of this thread have landed)
#ifdef TSAMD_SCHED_TIME
      if (n_sched >= 16u)
        printf("ts_schedule n=%u exchanges=%u | per SNP (us): head %.2f gamma %.2f first pass %.2f later passes %.2f tail %.2f | "
               "in exchanges %.2f, in folds %.2f, in epilogues %.2f, in sweeps %.2f | whole launch %.1f us\n", n_sched, xcount, tk_head * 0.01 / n_sched,
               tk_gamma * 0.01 / n_sched, tk_first * 0.01 / n_sched, tk_rest * 0.01 / n_sched, tk_tail * 0.01 / n_sched, tk_xchg * 0.01 / n_sched,
               tk_fold * 0.01 / n_sched, tk_epi * 0.01 / n_sched, tk_sweep * 0.01 / n_sched, (wall_clock64() - tk_start) * 0.01);
#endif
    }
    __syncthreads();
    if (p.host_error && tid < (uint32_t)TSAMD_PASS_HIST_BINS)  // the histogram's pinned host mirror (tsamd_pass_histogram reads it after a synchronise)
      __hip_atomic_store(p.host_error + 3 + tid, __hip_atomic_load(&ctl->pass_hist[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
#undef TSAMD_TK
}

}  // namespace tsamd
