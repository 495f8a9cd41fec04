// One instantiation unit of the K-specialised kernels: compiled once per K with
// -DTSAMD_K=<k> (terastructure_amd/build.py), so the builds run in parallel and the
// kernels see K as a compile-time constant.
#include "tsamd_resident_kernels.h"

#ifndef TSAMD_K
#error "compile with -DTSAMD_K=<populations>"
#endif

// leading scalar arguments of ts_pass (kernel-argument preload, tsamd_kernels.h), then the full parameter block
#define TSAMD_PASS_ARGS(chunk) \
  p.ctl, p.partials, p.w, p.npad, (chunk), par, nrows_hint, (p.xchg_world == 0u && p.rows_from_lt == 0u) ? 1u : 0u, p

#define TSAMD_CAT2(a, b) a##b
#define TSAMD_CAT(a, b) TSAMD_CAT2(a, b)

namespace tsamd {

// (kLaunchResident: `block` carries the chunk -- items per workgroup -- and `serial` the host's launch serial)
void TSAMD_CAT(launch_k, TSAMD_K)(int which, uint32_t grid, uint32_t block, hipStream_t stream, const DevParams &p,
                                  uint32_t par, uint32_t nrows_hint, uint32_t serial) {
  constexpr int K = TSAMD_K;
  switch (which) {
    case kLaunchPass:
      if (block == 1024)
        hipLaunchKernelGGL((ts_pass<K, false, 1024, 2>), dim3(grid), dim3(1024), 0, stream, TSAMD_PASS_ARGS(p.chunk));
      else if (block == 512)
        hipLaunchKernelGGL((ts_pass<K, false, 512, 2>), dim3(grid), dim3(512), 0, stream, TSAMD_PASS_ARGS(p.chunk));
      else
        hipLaunchKernelGGL((ts_pass<K, false, 256, 2>), dim3(grid), dim3(256), 0, stream, TSAMD_PASS_ARGS(p.chunk));
      break;
    case kLaunchFirst:
      if (block == 1)  // TSAMD_FIRST_VEC=2: two individuals per thread
        hipLaunchKernelGGL((ts_pass<K, true, 256, 2>), dim3(grid), dim3(256), 0, stream, TSAMD_PASS_ARGS(p.chunk_first));
      else
        hipLaunchKernelGGL((ts_pass<K, true, 256, 1>), dim3(grid), dim3(256), 0, stream, TSAMD_PASS_ARGS(p.chunk_first));
      break;
    case kLaunchResident:
      if constexpr (K <= kResidentMaxK)
        hipLaunchKernelGGL((ts_resident<K>), dim3(grid), dim3(kResidentBlock), 0, stream, p.ctl, p.partials, p.w, p.npad, block, par,
                           nrows_hint, p.res, serial, p);
      break;
    default:
      hipLaunchKernelGGL((ts_refresh_w<K>), dim3((p.npairs + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, p);
      break;
  }
}

// can a workgroup of the resident plain-pass kernel run on a compute unit (register budget)?
int TSAMD_CAT(resident_blocks_per_cu_k, TSAMD_K)() {
  constexpr int K = TSAMD_K;
  if constexpr (K <= kResidentMaxK) {
    int nb = 0;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ts_resident<K>, kResidentBlock, 0) == hipSuccess ? nb : 0;
  } else {
    return 0;
  }
}

// resident first-pass workgroups per compute unit (register-bound: 2 at K = 8, 1 from K = 12)
int TSAMD_CAT(first_blocks_per_cu_k, TSAMD_K)(int vec) {
  constexpr int K = TSAMD_K;
  int nb = 0;
  const hipError_t e = vec == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ts_pass<K, true, 256, 2>, 256, 0)
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ts_pass<K, true, 256, 1>, 256, 0);
  return e == hipSuccess ? nb : 0;
}

}  // namespace tsamd
