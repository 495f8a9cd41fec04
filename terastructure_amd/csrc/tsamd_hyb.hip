// The above-capacity whole-schedule kernel ts_hybrid<K, WR> (tsamd_hybrid_kernels.h), one translation unit per
// K <= kResidentMaxK, compiled with -DTSAMD_K=<k> and `-mllvm -disable-machine-licm` (terastructure_amd/build.py).
#include "tsamd_hybrid_kernels.h"

#ifndef TSAMD_K
#error "compile with -DTSAMD_K=<populations>"
#endif
#define TSAMD_CAT2(a, b) a##b
#define TSAMD_CAT(a, b) TSAMD_CAT2(a, b)

namespace tsamd {

static_assert(TSAMD_K <= kResidentMaxK, "ts_hybrid holds part of the shard's weights in registers");

// n entries at `sched` (pinned host or device memory), starting from and leaving the State of parity par; one GPU, or
// one launch per rank of a sharded run of up to 4 ranks (level 2 of the exchange spans the ranks' group leaders)
#define TSAMD_HYB_LAUNCH1(WR, STREAM)                                                                                                            \
  hipLaunchKernelGGL((ts_hybrid<TSAMD_K, WR, STREAM>), dim3(grid), dim3(kResidentBlock), 0, stream, p.ctl, p.w, p.npad, chunk, par, sched, n, p.res, \
                     serial, p)
// (a chunk that fits registers + LDS runs the instantiation without the streamed items' code)
#define TSAMD_HYB_LAUNCH(WR)                                      \
  do {                                                            \
    if (chunk > (uint32_t)hybrid_resident_capacity(TSAMD_K))      \
      TSAMD_HYB_LAUNCH1(WR, true);                                \
    else                                                          \
      TSAMD_HYB_LAUNCH1(WR, false);                               \
  } while (0)
void TSAMD_CAT(launch_hybrid_k, TSAMD_K)(uint32_t grid, uint32_t chunk, hipStream_t stream, const DevParams &p, uint32_t par,
                                         const uint32_t *sched, uint32_t n, uint32_t serial) {
  const uint32_t world = p.xchg_world;
  if (world == 0u)
    TSAMD_HYB_LAUNCH(0);
  else if (world <= 2u)
    TSAMD_HYB_LAUNCH(8);
  else
    TSAMD_HYB_LAUNCH(16);
}

// does a workgroup of it fit a compute unit (register / LDS budget)?  (worst case of the instantiations)
int TSAMD_CAT(hybrid_blocks_per_cu_k, TSAMD_K)() {
  int worst = 1 << 30;
  auto probe = [&](auto kernel) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kResidentBlock, 0) != hipSuccess) nb = 0;
    worst = nb < worst ? nb : worst;
  };
  probe(ts_hybrid<TSAMD_K, 0, true>);
  probe(ts_hybrid<TSAMD_K, 8, true>);
  probe(ts_hybrid<TSAMD_K, 16, true>);
  probe(ts_hybrid<TSAMD_K, 0, false>);
  probe(ts_hybrid<TSAMD_K, 8, false>);
  probe(ts_hybrid<TSAMD_K, 16, false>);
  return worst;
}

// individuals per workgroup whose weights never leave the chip (registers + LDS)
int TSAMD_CAT(hybrid_capacity_k, TSAMD_K)() { return hybrid_resident_capacity(TSAMD_K); }

}  // namespace tsamd
