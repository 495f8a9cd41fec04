// The batched validation-mode kernel for shards above the register capacity, ts_hybhol<K> (tsamd_hybhol_kernels.h), one
// translation unit per K <= kResidentMaxK, compiled with -DTSAMD_K=<k> (terastructure_amd/build.py).
#include "tsamd_hybhol_kernels.h"

#ifndef TSAMD_K
#error "compile with -DTSAMD_K=<populations>"
#endif
#define TSAMD_CAT2(a, b) a##b
#define TSAMD_CAT(a, b) TSAMD_CAT2(a, b)

namespace tsamd {

static_assert(TSAMD_K <= kResidentMaxK, "ts_hybhol is specialised per K");
static_assert(hh_reg_items(TSAMD_K) + hh_lds_items(TSAMD_K) >= 1, "at least one item on chip");

// n hol-mode entries at `sched` (pinned host memory), pairwise distinct locations, no gamma step pending; ts_hybrid's launch
// geometry (its per-thread partial sums are the same sums, whatever holds the weights)
void TSAMD_CAT(launch_hybhol_k, TSAMD_K)(uint32_t grid, uint32_t chunk, hipStream_t stream, const DevParams &p, uint32_t par,
                                         const uint32_t *sched, uint32_t n, uint32_t serial) {
  // (a sharded context -- ts_hybrid runs on up to 4 ranks: level 2 of the exchanges spans the ranks' group leaders)
  if (p.xchg_world == 0u)
    hipLaunchKernelGGL((ts_hybhol<TSAMD_K, 0>), dim3(grid), dim3(kResidentBlock), 0, stream, p.ctl, p.w, p.npad, chunk, par, sched, n, p.res, serial, p);
  else if (p.xchg_world <= 2u)
    hipLaunchKernelGGL((ts_hybhol<TSAMD_K, 8>), dim3(grid), dim3(kResidentBlock), 0, stream, p.ctl, p.w, p.npad, chunk, par, sched, n, p.res, serial, p);
  else
    hipLaunchKernelGGL((ts_hybhol<TSAMD_K, 16>), dim3(grid), dim3(kResidentBlock), 0, stream, p.ctl, p.w, p.npad, chunk, par, sched, n, p.res, serial, p);
}

int TSAMD_CAT(hybhol_blocks_per_cu_k, TSAMD_K)() {
  int nb = 0, nb2 = 0, nb3 = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ts_hybhol<TSAMD_K, 0>, kResidentBlock, 0) != hipSuccess) nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb2, ts_hybhol<TSAMD_K, 8>, kResidentBlock, 0) != hipSuccess) nb2 = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb3, ts_hybhol<TSAMD_K, 16>, kResidentBlock, 0) != hipSuccess) nb3 = 0;
  return nb < nb2 ? (nb < nb3 ? nb : nb3) : (nb2 < nb3 ? nb2 : nb3);
}

// locations per exchange (what tsamd_holblock_info reports)
int TSAMD_CAT(hybhol_batch_k, TSAMD_K)() { return hh_batch(TSAMD_K); }

}  // namespace tsamd
