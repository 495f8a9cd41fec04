// The batched validation-mode kernel ts_holblock<K> (tsamd_holblock_kernels.h), one translation unit per
// K <= kResidentMaxK, compiled with -DTSAMD_K=<k> (terastructure_amd/build.py).
#include "tsamd_holblock_kernels.h"

#ifndef TSAMD_K
#error "compile with -DTSAMD_K=<populations>"
#endif
#define TSAMD_CAT2(a, b) a##b
#define TSAMD_CAT(a, b) TSAMD_CAT2(a, b)

namespace tsamd {

static_assert(TSAMD_K <= kResidentMaxK, "ts_holblock holds the shard's weights in registers");

// n hol-mode entries at `sched` (pinned host memory), pairwise distinct locations, no gamma step pending; same launch
// geometry as ts_schedule (its per-thread partial sums are the same sums)
void TSAMD_CAT(launch_holblock_k, TSAMD_K)(uint32_t grid, uint32_t chunk, hipStream_t stream, const DevParams &p, uint32_t par,
                                           const uint32_t *sched, uint32_t n, uint32_t serial) {
  // (a sharded context: the instantiation whose level 2 spans the ranks' group leaders -- WR = 8 up to 2 ranks; 32 above: up to 64
  // rows, polled 16 row pairs at a time)
  if (p.xchg_world == 0u)
    hipLaunchKernelGGL((ts_holblock<TSAMD_K, 0>), dim3(grid), dim3(kResidentBlock), 0, stream, p.ctl, p.w, p.npad, chunk, par, sched, n, p.res, serial, p);
  else if (p.xchg_world <= 2u)
    hipLaunchKernelGGL((ts_holblock<TSAMD_K, 8>), dim3(grid), dim3(kResidentBlock), 0, stream, p.ctl, p.w, p.npad, chunk, par, sched, n, p.res, serial, p);
  else
    hipLaunchKernelGGL((ts_holblock<TSAMD_K, 32>), dim3(grid), dim3(kResidentBlock), 0, stream, p.ctl, p.w, p.npad, chunk, par, sched, n, p.res, serial,
                       p);
}

int TSAMD_CAT(holblock_blocks_per_cu_k, TSAMD_K)() {
  int nb = 0, nb2 = 0, nb3 = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ts_holblock<TSAMD_K, 0>, kResidentBlock, 0) != hipSuccess) nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb2, ts_holblock<TSAMD_K, 8>, kResidentBlock, 0) != hipSuccess) nb2 = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb3, ts_holblock<TSAMD_K, 32>, kResidentBlock, 0) != hipSuccess) nb3 = 0;
  return nb < nb2 ? (nb < nb3 ? nb : nb3) : (nb2 < nb3 ? nb2 : nb3);
}

// locations per exchange / per launch (what the host cuts a validation-mode schedule into)
int TSAMD_CAT(holblock_batch_k, TSAMD_K)() { return hol_batch(TSAMD_K); }

}  // namespace tsamd
