// The whole-schedule kernel ts_schedule<K> (tsamd_resident_kernels.h), one translation unit per K <= kResidentMaxK,
// compiled with -DTSAMD_K=<k> and `-mllvm -disable-machine-licm` (terastructure_amd/build.py): the kernel
// is one loop over the schedule around fully unrolled sweeps, and loop-invariant code motion would hoist
// a few hundred addresses and constants out of that loop into registers the kernel needs for the weights.
#include "tsamd_resident_kernels.h"

#ifndef TSAMD_K
#error "compile with -DTSAMD_K=<populations>"
#endif
#define TSAMD_CAT2(a, b) a##b
#define TSAMD_CAT(a, b) TSAMD_CAT2(a, b)

namespace tsamd {

static_assert(TSAMD_K <= kResidentMaxK, "ts_schedule holds the shard's weights in registers");
#ifndef TSAMD_ALWAYS_PARTIAL  // (experiment: 1 = every K runs the instantiation with the skip branches)
#define TSAMD_ALWAYS_PARTIAL 0
#endif
constexpr bool kAlwaysPartial = (TSAMD_K) > 8 || TSAMD_ALWAYS_PARTIAL;
static_assert(offsetof(ResXchg, abort_word) == 0, "sequence_aborted() reads the first word of the buffer");
static_assert(sizeof(((Xchg *)nullptr)->res_sums) / sizeof(unsigned long long) >= 2u * 2u * kMaxRanks * kResGroups * ResLay<TSAMD_K>::GR,
              "Xchg::res_sums holds two slots x two regions of world x 8 rows");

// n entries at `sched` (pinned host or device memory), starting from and leaving the State of parity par.
// A chunk that fills all items of its threads runs the kernel without the skip-unused-items branches; a sharded
// context (p.xchg_world ranks connected peer to peer) runs the instantiation whose level 2 spans the ranks' group
// leaders: WR = 8 rows per half for up to 2 ranks, 16 for up to 4, 32 for up to 8 -- up to K = 16 the two halves of the
// world x 8 rows are polled by two waves side by side (WR / 2 row pairs per lane), above by one wave per column block
// (res_exchange, tsamd_resident_kernels.h).
#define TSAMD_SCHED_LAUNCH(PARTIAL, WR)                                                                                               \
  hipLaunchKernelGGL((ts_schedule<TSAMD_K, (PARTIAL) || kAlwaysPartial, WR>), dim3(grid), dim3(kResidentBlock), 0, stream, p.ctl, p.w, p.npad, \
                     chunk, par, sched, n, p.res, serial, p)
void TSAMD_CAT(launch_schedule_k, TSAMD_K)(uint32_t grid, uint32_t chunk, hipStream_t stream, const DevParams &p, uint32_t par,
                                           const uint32_t *sched, uint32_t n, uint32_t serial) {
  // (K > 8: always the instantiation with the skip branches -- without them the scheduler moves the items' loads so far
  // ahead that the kernel no longer fits the register file)
  const bool partial = kAlwaysPartial || chunk <= (uint32_t)((resident_items(TSAMD_K) - 1) * kResidentBlock);
  const uint32_t world = p.xchg_world;
  if (world == 0u) {
    if (partial) TSAMD_SCHED_LAUNCH(true, 0); else TSAMD_SCHED_LAUNCH(false, 0);
  } else if (world <= 2u) {
    if (partial) TSAMD_SCHED_LAUNCH(true, 8); else TSAMD_SCHED_LAUNCH(false, 8);
  } else if (world <= 4u) {
    if (partial) TSAMD_SCHED_LAUNCH(true, 16); else TSAMD_SCHED_LAUNCH(false, 16);
  } else {
    if (partial) TSAMD_SCHED_LAUNCH(true, 32); else TSAMD_SCHED_LAUNCH(false, 32);
  }
}

// does a workgroup of it fit a compute unit (register budget)?  (worst case of the instantiations)
int TSAMD_CAT(schedule_blocks_per_cu_k, TSAMD_K)() {
  int worst = 1 << 30;
  auto probe = [&](auto kernel) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kResidentBlock, 0) != hipSuccess) nb = 0;
    worst = nb < worst ? nb : worst;
  };
  probe(ts_schedule<TSAMD_K, kAlwaysPartial, 0>);
  probe(ts_schedule<TSAMD_K, true, 0>);
  probe(ts_schedule<TSAMD_K, kAlwaysPartial, 8>);
  probe(ts_schedule<TSAMD_K, true, 8>);
  probe(ts_schedule<TSAMD_K, kAlwaysPartial, 16>);
  probe(ts_schedule<TSAMD_K, true, 16>);
  probe(ts_schedule<TSAMD_K, kAlwaysPartial, 32>);
  probe(ts_schedule<TSAMD_K, true, 32>);
  return worst;
}

}  // namespace tsamd
