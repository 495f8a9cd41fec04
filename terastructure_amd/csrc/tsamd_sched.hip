// The whole-schedule kernel ts_schedule<K> (tsamd_kernels.h), one translation unit per K <= kResidentMaxK,
// compiled with -DTSAMD_K=<k> and `-mllvm -disable-machine-licm` (terastructure_amd/build.py): the kernel
// is one loop over the schedule around fully unrolled sweeps, and loop-invariant code motion would hoist
// a few hundred addresses and constants out of that loop into registers the kernel needs for the weights.
#include "tsamd_kernels.h"

#ifndef TSAMD_K
#error "compile with -DTSAMD_K=<populations>"
#endif
#define TSAMD_CAT2(a, b) a##b
#define TSAMD_CAT(a, b) TSAMD_CAT2(a, b)

namespace tsamd {

static_assert(TSAMD_K <= kResidentMaxK, "ts_schedule holds the shard's weights in registers: K <= 8");

// n entries at `sched` (device memory), starting from and leaving the State of parity par
void TSAMD_CAT(launch_schedule_k, TSAMD_K)(uint32_t grid, hipStream_t stream, const DevParams &p, uint32_t par, const uint32_t *sched,
                                           uint32_t n) {
  // a chunk that fills all eight items of its threads runs the kernel without the skip-unused-items branches
  if (p.chunk > (uint32_t)((kResidentItems - 1) * kResidentBlock))
    hipLaunchKernelGGL((ts_schedule<TSAMD_K, false>), dim3(grid), dim3(kResidentBlock), 0, stream, p.ctl, p.w, p.npad, p.chunk, par,
                       sched, n, p.res, p);
  else
    hipLaunchKernelGGL((ts_schedule<TSAMD_K, true>), dim3(grid), dim3(kResidentBlock), 0, stream, p.ctl, p.w, p.npad, p.chunk, par,
                       sched, n, p.res, p);
}

// does a workgroup of it fit a compute unit (register budget)?
int TSAMD_CAT(schedule_blocks_per_cu_k, TSAMD_K)() {
  int nb = 0, nb2 = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ts_schedule<TSAMD_K, false>, kResidentBlock, 0) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb2, ts_schedule<TSAMD_K, true>, kResidentBlock, 0) != hipSuccess) return 0;
  return nb < nb2 ? nb : nb2;
}

}  // namespace tsamd
