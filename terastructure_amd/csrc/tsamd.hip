// libtsamd.so -- C ABI (include/tsamd.h) over the HIP kernels in tsamd_kernels.h.
// gfx950 only.  No CPU fallback anywhere: every entry point drives the device.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_set>
#include <vector>

#include "tsamd.h"
#define TSAMD_MAIN_TU 1
#include "tsamd_generic_kernels.h"
#include "tsamd_kernels.h"
#include "tsamd_resident_kernels.h"
#include "tsamd_holblock_kernels.h"
#include "tsamd_hybrid_kernels.h"
#include "tsamd_wide_kernels.h"

using namespace tsamd;

namespace {

std::string g_create_error;

struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
  bool load() {
    if (handle) return true;
    // RTLD_NOLOAD first: inside a torch process reuse the RCCL torch already mapped.
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *nm : names) {
      handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
      if (handle) break;
    }
    for (const char *nm : names) {
      if (handle) break;
      handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!handle) {
      error = std::string("dlopen librccl failed: ") + dlerror();
      return false;
    }
    GetUniqueId = (decltype(GetUniqueId))dlsym(handle, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(handle, "ncclCommInitRank");
    AllReduce = (decltype(AllReduce))dlsym(handle, "ncclAllReduce");
    CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
    GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy || !GetErrorString) {
      error = "librccl is missing a required symbol";
      return false;
    }
    return true;
  }
};
RcclApi g_rccl;

struct HeldLoc {
  std::vector<uint32_t> local_ids;  // ascending
  std::vector<uint8_t> ytrue;       // 0/1/2
};

constexpr uint32_t kProfCap = 8192;
constexpr uint32_t kGraphLevels = 5;                     // captured graphs of 1, 2, 4, 8 and 16 SNPs ...
constexpr uint32_t kGraphSnps = 1u << (kGraphLevels - 1);  // ... so that any schedule length replays without padding

struct GraphSlot {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

}  // namespace

struct tsamd_ctx {
  tsamd_config cfg{};
  int dev = 0;
  hipStream_t stream = nullptr;
  uint32_t n_begin = 0, n_local = 0, npad = 0;
  bool wide = false;  // K above TSAMD_SPECIALIZED_K: run-time-K fallback kernels (tsamd_wide_kernels.h)
  uint32_t grid = 0, block = 256, grid_first = 0, first_vec = 1;  // plain-pass and first-pass launch geometry
  DevParams p{};
  uint32_t *d_sched = nullptr;
  uint32_t sched_cap = 0;
  uint8_t *h_stage = nullptr;  // pinned staging for uploads
  size_t stage_bytes = 0;
  std::map<uint32_t, HeldLoc> held;
  // flat device copy of the held-out table (ids + true genotypes, locations ascending), rebuilt
  // lazily after tsamd_set_heldout / a re-upload; spans index it by location
  bool held_dirty = true;
  std::map<uint32_t, std::pair<uint64_t, uint32_t>> held_span;
  uint32_t *d_hids = nullptr;
  uint8_t *d_hy = nullptr;
  double *d_hterms = nullptr;
  size_t held_cap = 0;
  uint32_t *d_fold_ids = nullptr;  // scratch of tsamd_set_heldout
  uint8_t *d_fold_orig = nullptr;
  size_t fold_cap = 0;
  HeldReq *d_hreq = nullptr;
  double *d_hsums = nullptr;
  size_t hreq_cap = 0;
  ncclComm_t comm = nullptr;
  Xchg *xchg = nullptr;                    // peer-to-peer exchange buffer (fine-grained, IPC-exported)
  std::vector<void *> peer_maps;           // hipIpcOpenMemHandle results to close
  bool p2p = false;
  bool split = false;  // lambda_t leaves the pass via ctl->lt and the epilogue is its own kernel
  bool rccl_graph = false;  // TSAMD_RCCL_GRAPH=1: capture the RCCL all-reduce into the replayed graphs
  bool resident = false;    // plain passes of a SNP run as ONE launch (ts_resident) instead of max_inner - 1
  bool persistent = false;  // ... and a whole schedule runs as ONE launch (ts_schedule: the weights never leave the registers)
  bool can_resident = false, can_persistent = false;  // what the context qualifies for (tsamd_set_launch_mode)
  bool hybrid = false;             // the whole-schedule kernel of this context is ts_hybrid: the shard exceeds ts_schedule's register capacity
  bool can_holblock = false;       // ... and validation-mode schedules run batched (ts_holblock) while it runs ts_schedule
  bool can_hybhol = false;         // ... or (one GPU) batched by ts_hybhol while it runs ts_hybrid
  bool tail_step_pending = false;  // the last entry enqueued was a training update: its gamma step is pending
  uint64_t holblock_launches = 0, holblock_locs = 0;
  uint32_t sched_grid = 0, sched_chunk = 0;  // launch geometry of ts_schedule (= the plain pass' on one GPU; its own when sharded)
  uint32_t res_grid = 0, res_chunk = 0;      // ... and of ts_resident: the same shard, shrunk only as far as ITS exchange has one level
  uint32_t device_share = 1;  // contexts whose resident kernels share this device (tests: several ranks on one GPU)
  ResXchg *res = nullptr;   // their in-launch exchange buffer
  unsigned long long *h_error = nullptr;  // pinned: tag of a bounded in-kernel wait that gave up (0: none)
  // profiling
  bool prof = false;
  std::vector<hipEvent_t> ev_pass, ev_first;  // start/stop pairs
  uint32_t n_ev_pass = 0, n_ev_first = 0;
  uint64_t prof_pass_n = 0, prof_first_n = 0;
  uint64_t prof_passes0 = 0;  // ctl->total_passes when profiling was enabled
  bool prof_capped = false;   // more SNPs than event pairs: not every bracket was timed
  double prof_pass_ms = 0, prof_first_ms = 0;
  // hipGraph replay of the per-SNP kernel sequence: [log2 SNPs][launch parity on entry]
  GraphSlot graphs[kGraphLevels][2];
  bool graphs_ready = false;
  uint64_t q = 0;  // kernels of the state-machine sequence launched so far (parity = q & 1)
  uint32_t prev_rows = 0;  // grid of the last pass kernel enqueued (row-count hint for the next)
  // Pinned host copies of the schedules enqueued since the stream was last known idle, in order (the kernels read
  // them; recycled by settle()).  Also the journal a failed resident launch is replayed from: mode = how the entry was
  // launched (0 per pass, 1 per SNP, 2 per schedule), serial0 = launch serial of its first resident launch.
  struct Journal {
    uint32_t *ent;
    size_t cap;
    uint32_t n, serial0;
    int mode;
    std::vector<uint32_t> launch_off;  // mode 2: first entry of each of its resident launches (ts_schedule / ts_holblock), in serial order
    hipEvent_t done;                   // peer-to-peer contexts: recorded behind the entry's last kernel (the journal is trimmed without a synchronise)
  };
  std::vector<Journal> journal;
  std::vector<std::pair<uint32_t *, size_t>> sched_free;
  std::vector<hipEvent_t> event_free;
  uint32_t launch_serial = 0;   // resident launches so far (ts_resident / ts_schedule carry it; a failing one reports it)
  bool recovering = false;
  uint32_t recoveries = 0;      // times a resident launch gave up at its entry and the schedule was replayed launch per pass
  hipStream_t aux_stream = nullptr;      // tsamd_debug_occupy
  unsigned long long *h_occupy = nullptr;  // pinned: its kernel has started
  std::string err;
};

namespace {

int fail(tsamd_ctx *ctx, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx)
    ctx->err = buf;
  else
    g_create_error = buf;
  return code;
}

#define HIP_TRY(ctx, expr)                                                                     \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(ctx, e_ == hipErrorOutOfMemory ? TSAMD_ENOMEM : TSAMD_EHIP, "%s: %s", #expr, \
                  hipGetErrorString(e_));                                                      \
  } while (0)

#define CHECK_CTX(ctx) \
  if (!(ctx)) return TSAMD_EINVAL

// launchers of the K-specialised kernels, one per translation unit (tsamd_inst.hip)
#define TSAMD_DECL(k)                                                                                        \
  void launch_k##k(int, uint32_t, uint32_t, hipStream_t, const DevParams &, uint32_t, uint32_t, uint32_t); \
  int first_blocks_per_cu_k##k(int);                                                                         \
  int resident_blocks_per_cu_k##k();                                                                         \
  void launch_schedule_k##k(uint32_t, uint32_t, hipStream_t, const DevParams &, uint32_t, const uint32_t *, uint32_t, uint32_t); \
  int schedule_blocks_per_cu_k##k();                                                                         \
  void launch_holblock_k##k(uint32_t, uint32_t, hipStream_t, const DevParams &, uint32_t, const uint32_t *, uint32_t, uint32_t); \
  int holblock_blocks_per_cu_k##k();                                                                         \
  void launch_hybrid_k##k(uint32_t, uint32_t, hipStream_t, const DevParams &, uint32_t, const uint32_t *, uint32_t, uint32_t); \
  int hybrid_blocks_per_cu_k##k();                                                                           \
  void launch_hybhol_k##k(uint32_t, uint32_t, hipStream_t, const DevParams &, uint32_t, const uint32_t *, uint32_t, uint32_t); \
  int hybhol_blocks_per_cu_k##k();                                                                           \
  int hybhol_batch_k##k();
#define TSAMD_ALL_K(X)                                                                             \
  X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17)     \
  X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)
}  // namespace
namespace tsamd {
TSAMD_ALL_K(TSAMD_DECL)  // tsamd_inst.hip, tsamd_sched.hip, tsamd_hol.hip, tsamd_hyb.hip and tsamd_hhol.hip: five translation units per K
}
namespace {
#define TSAMD_ENTRY(k) tsamd::launch_k##k,
const LaunchFn kLaunchers[TSAMD_SPECIALIZED_K + 1] = {nullptr, TSAMD_ALL_K(TSAMD_ENTRY)};
#define TSAMD_OCC_ENTRY(k) tsamd::first_blocks_per_cu_k##k,
int (*const kFirstBlocksPerCu[TSAMD_SPECIALIZED_K + 1])(int) = {nullptr, TSAMD_ALL_K(TSAMD_OCC_ENTRY)};
#define TSAMD_RES_ENTRY(k) tsamd::resident_blocks_per_cu_k##k,
int (*const kResidentBlocksPerCu[TSAMD_SPECIALIZED_K + 1])() = {nullptr, TSAMD_ALL_K(TSAMD_RES_ENTRY)};
static_assert(kResidentMaxK == TSAMD_SPECIALIZED_K, "TSAMD_ALL_K lists K = 1 .. kResidentMaxK");
typedef void (*ScheduleFn)(uint32_t, uint32_t, hipStream_t, const DevParams &, uint32_t, const uint32_t *, uint32_t, uint32_t);
#define TSAMD_SCHED_ENTRY(k) tsamd::launch_schedule_k##k,
const ScheduleFn kScheduleLaunchers[kResidentMaxK + 1] = {nullptr, TSAMD_ALL_K(TSAMD_SCHED_ENTRY)};
#define TSAMD_SCHED_OCC_ENTRY(k) tsamd::schedule_blocks_per_cu_k##k,
int (*const kScheduleBlocksPerCu[kResidentMaxK + 1])() = {nullptr, TSAMD_ALL_K(TSAMD_SCHED_OCC_ENTRY)};
#define TSAMD_HOL_ENTRY(k) tsamd::launch_holblock_k##k,
const ScheduleFn kHolblockLaunchers[kResidentMaxK + 1] = {nullptr, TSAMD_ALL_K(TSAMD_HOL_ENTRY)};
#define TSAMD_HOL_OCC_ENTRY(k) tsamd::holblock_blocks_per_cu_k##k,
int (*const kHolblockBlocksPerCu[kResidentMaxK + 1])() = {nullptr, TSAMD_ALL_K(TSAMD_HOL_OCC_ENTRY)};
#define TSAMD_HYB_ENTRY(k) tsamd::launch_hybrid_k##k,
const ScheduleFn kHybridLaunchers[kResidentMaxK + 1] = {nullptr, TSAMD_ALL_K(TSAMD_HYB_ENTRY)};
#define TSAMD_HYB_OCC_ENTRY(k) tsamd::hybrid_blocks_per_cu_k##k,
int (*const kHybridBlocksPerCu[kResidentMaxK + 1])() = {nullptr, TSAMD_ALL_K(TSAMD_HYB_OCC_ENTRY)};
#define TSAMD_HHOL_ENTRY(k) tsamd::launch_hybhol_k##k,
const ScheduleFn kHybholLaunchers[kResidentMaxK + 1] = {nullptr, TSAMD_ALL_K(TSAMD_HHOL_ENTRY)};
#define TSAMD_HHOL_OCC_ENTRY(k) tsamd::hybhol_blocks_per_cu_k##k,
int (*const kHybholBlocksPerCu[kResidentMaxK + 1])() = {nullptr, TSAMD_ALL_K(TSAMD_HHOL_OCC_ENTRY)};
#define TSAMD_HHOL_BATCH_ENTRY(k) tsamd::hybhol_batch_k##k,
int (*const kHybholBatch[kResidentMaxK + 1])() = {nullptr, TSAMD_ALL_K(TSAMD_HHOL_BATCH_ENTRY)};

__global__ void ts_fill_f64(double *p, size_t n, double v0, double v1) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (i & 1) ? v1 : v0;
}

// Every kernel of the state-machine sequence gets the next parity bit (tsamd_device.h).
uint32_t next_parity(tsamd_ctx *c) { return (uint32_t)(c->q++ & 1u); }

// one pass = ts_pass [+ ts_reduce_rows + all-reduce when sharded]; `pass` = its index within
// the SNP (0 = first pass).  Odd plain passes sweep backwards (bit 1 of the parity argument).
int enqueue_pass(tsamd_ctx *c, uint32_t pass) {
  const bool first = pass == 0;
  const uint32_t par = next_parity(c);
  const uint32_t par_arg = par | ((c->p.sweep_alternate ? (pass & 1u) : 0u) << 1);
  // rows of the previous launch of the sequence: a first pass follows a plain pass (or a
  // kernel that left nothing pending), a plain pass follows the first pass or a plain pass
  const uint32_t hint = c->prev_rows;
  if (c->wide) {
    if (first)
      hipLaunchKernelGGL((ts_pass_wide<true>), dim3(c->grid_first), dim3(kWideBlock), 0, c->stream, c->p, par_arg, hint);
    else
      hipLaunchKernelGGL((ts_pass_wide<false>), dim3(c->grid_first), dim3(kWideBlock), 0, c->stream, c->p, par_arg, hint);
    c->prev_rows = c->grid_first;
  } else if (first)
    kLaunchers[c->cfg.k](kLaunchFirst, c->grid_first, c->first_vec == 2 ? 1u : 0u, c->stream, c->p, par_arg, hint, 0u);
  else
    kLaunchers[c->cfg.k](kLaunchPass, c->grid, c->block, c->stream, c->p, par_arg, hint, 0u);
  if (!c->wide) c->prev_rows = first ? c->grid_first : c->grid;
  if (c->split && !c->p2p) {  // (peer-to-peer: every workgroup has already pushed its row to every rank)
    hipLaunchKernelGGL(ts_reduce_rows, dim3(1), dim3(256), 0, c->stream, c->p, par);
    Ctl *ctl = c->p.ctl;
    if (c->comm) {
      ncclResult_t r = g_rccl.AllReduce(ctl->lt[par], ctl->lt_sum[par], 2 * c->cfg.k, ncclDouble, ncclSum, c->comm,
                                        c->stream);
      if (r != ncclSuccess) return fail(c, TSAMD_ECOMM, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
    } else {
      HIP_TRY(c, hipMemcpyAsync(ctl->lt_sum[par], ctl->lt[par], sizeof(double) * 2 * c->cfg.k,
                                hipMemcpyDeviceToDevice, c->stream));
    }
  }
  return TSAMD_OK;
}

void enqueue_begin(tsamd_ctx *c, uint32_t n, bool drop_pending, const uint32_t *host_sched = nullptr) {
  hipLaunchKernelGGL(ts_begin, dim3(1), dim3(256), 0, c->stream, c->p, host_sched, c->d_sched, n, next_parity(c),
                     drop_pending ? 1u : 0u);
}

void enqueue_flush(tsamd_ctx *c) {
  if (c->wide)
    hipLaunchKernelGGL((ts_flush<kWideBlock>), dim3(1), dim3(kWideBlock), 0, c->stream, c->p, next_parity(c));
  else  // 256 = workgroup size of ts_pass<K, true, ...>
    hipLaunchKernelGGL((ts_flush<256>), dim3(1), dim3(256), 0, c->stream, c->p, next_parity(c));
}

// profiling: one HIP-event pair around the first pass and one around the run of plain
// passes of each SNP (a bracket per launch would mostly time the dispatch gap of an
// eagerly launched ~10 us kernel)
int prof_event(tsamd_ctx *c, std::vector<hipEvent_t> &evs, uint32_t slot, hipEvent_t *out) {
  while (evs.size() <= slot) {
    hipEvent_t e;
    HIP_TRY(c, hipEventCreate(&e));
    evs.push_back(e);
  }
  *out = evs[slot];
  return TSAMD_OK;
}

int enqueue_snp(tsamd_ctx *c) {
  const bool prof = c->prof && c->n_ev_first < kProfCap;
  if (c->prof && !prof) c->prof_capped = true;
  hipEvent_t e = nullptr;
  if (prof) {
    if (int rc = prof_event(c, c->ev_first, 2 * c->n_ev_first, &e)) return rc;
    HIP_TRY(c, hipEventRecord(e, c->stream));
  }
  int rc = enqueue_pass(c, 0);
  if (prof) {
    if (int rc2 = prof_event(c, c->ev_first, 2 * c->n_ev_first + 1, &e)) return rc2;
    HIP_TRY(c, hipEventRecord(e, c->stream));
    c->n_ev_first++;
    if (c->cfg.max_inner > 1) {
      if (int rc2 = prof_event(c, c->ev_pass, 2 * c->n_ev_pass, &e)) return rc2;
      HIP_TRY(c, hipEventRecord(e, c->stream));
    }
  }
  if (c->resident) {  // every plain pass of the SNP in one launch
    if (rc == TSAMD_OK) {
      const uint32_t par = next_parity(c);
      kLaunchers[c->cfg.k](kLaunchResident, c->res_grid, c->res_chunk, c->stream, c->p, par, c->prev_rows, c->launch_serial++);
      c->prev_rows = c->res_grid;
    }
  } else {
    for (uint32_t i = 1; rc == TSAMD_OK && i < c->cfg.max_inner; ++i) rc = enqueue_pass(c, i);
  }
  if (prof && c->cfg.max_inner > 1 && rc == TSAMD_OK) {
    if (int rc2 = prof_event(c, c->ev_pass, 2 * c->n_ev_pass + 1, &e)) return rc2;
    HIP_TRY(c, hipEventRecord(e, c->stream));
    c->n_ev_pass++;
  }
  return rc;
}

void destroy_graph(tsamd_ctx *c) {
  for (auto &lvl : c->graphs)
    for (GraphSlot &g : lvl) {
      if (g.exec) hipGraphExecDestroy(g.exec);
      if (g.graph) hipGraphDestroy(g.graph);
      g.exec = nullptr;
      g.graph = nullptr;
    }
  c->graphs_ready = false;
}

// kernels of the state-machine sequence per SNP (ts_reduce_rows shares its pass' parity)
uint32_t kernels_per_snp(const tsamd_ctx *c) { return c->resident ? 2u : c->cfg.max_inner; }
// SNPs per ts_schedule launch at most (a launch of this many runs for seconds; every in-kernel wait is bounded)
constexpr uint32_t kScheduleChunk = 1u << 16;

// Capture 2^level consecutive SNP sequences for launch parity par0 on entry.  Kernels read
// everything that varies (location, pending state) from device memory; the only frozen
// arguments are the parity / sweep-direction bits, which is why a graph exists per entry parity.
int build_graph(tsamd_ctx *c, uint32_t level, uint32_t par0) {
  GraphSlot &slot = c->graphs[level][par0];
  const uint64_t q_save = c->q;
  const uint32_t rows_save = c->prev_rows;
  c->q = par0;
  // in replay the first kernel follows the last kernel of a SNP sequence (or nothing pending)
  c->prev_rows = c->cfg.max_inner > 1 ? c->grid : c->grid_first;
  HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  int rc = TSAMD_OK;
  for (uint32_t s = 0; s < (1u << level) && rc == TSAMD_OK; ++s) rc = enqueue_snp(c);
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(c->stream, &g);
  c->q = q_save;
  c->prev_rows = rows_save;
  if (rc != TSAMD_OK) {
    if (g) hipGraphDestroy(g);
    return rc;
  }
  if (e != hipSuccess) return fail(c, TSAMD_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
  slot.graph = g;
  HIP_TRY(c, hipGraphInstantiate(&slot.exec, slot.graph, nullptr, nullptr, 0));
  if (hipGraphUpload(slot.exec, c->stream) != hipSuccess) (void)hipGetLastError();  // (optional in this runtime)
  return TSAMD_OK;
}

// All graphs at once (10 captures, 62 SNP sequences): whatever the first schedule's length is,
// no later call pays for a capture or an instantiation.
int build_all_graphs(tsamd_ctx *c) {
  if (c->graphs_ready) return TSAMD_OK;
  for (uint32_t level = 0; level < kGraphLevels; ++level)
    for (uint32_t par0 = 0; par0 < 2; ++par0)
      if (!c->graphs[level][par0].exec)
        if (int rc = build_graph(c, level, par0)) return rc;
  c->graphs_ready = true;
  return TSAMD_OK;
}

bool graphs_allowed(const tsamd_ctx *c) {
  // (RCCL all-reduce inside the captured sequence: opt-in, TSAMD_RCCL_GRAPH=1 on every rank)
  const bool comm_graph = c->comm && !c->p2p && c->rccl_graph;
  // (resident plain passes: two launches per SNP of ~50 us each -- eager launches stay far ahead of the
  // device and have neither the submission cost of a graph nor the ~8 us boundary between graphs)
  return !(c->cfg.flags & TSAMD_FLAG_NO_GRAPH) && (!c->comm || c->p2p || comm_graph) && !c->prof && !c->resident;
}

// the graphs are built on first use; with an RCCL all-reduce inside, the first collective runs
// outside the capture (RCCL sets its channels up lazily)
int ensure_graphs(tsamd_ctx *c) {
  if (c->graphs_ready) return TSAMD_OK;
  if (c->comm && !c->p2p && c->rccl_graph) {
    ncclResult_t r = g_rccl.AllReduce(c->p.ctl->lt[0], c->p.ctl->lt_sum[0], 2 * c->cfg.k, ncclDouble, ncclSum, c->comm,
                                      c->stream);
    if (r != ncclSuccess) return fail(c, TSAMD_ECOMM, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  return build_all_graphs(c);
}

uint32_t env_u32(const char *name, uint32_t dflt) {
  const char *s = getenv(name);
  return (s && *s) ? (uint32_t)std::max(0, atoi(s)) : dflt;
}

// Launch geometry.  The pass kernel is a streaming reduction: enough waves per CU to cover
// HBM latency, but few workgroups, because every workgroup of the NEXT launch adds all
// partial rows up again (and, sharded peer-to-peer, every workgroup sends its row to every
// rank: max_grid = kXchgBlocks there).
void configure_launch(tsamd_ctx *c, uint32_t max_grid) {
  DevParams &p = c->p;
  if (c->wide) {  // one individual per thread, at most kWideItems individuals per thread
    uint32_t chunk = (p.npad + max_grid - 1) / max_grid;
    chunk = (chunk + kWideBlock - 1) / kWideBlock * kWideBlock;
    p.chunk_first = p.chunk = chunk;
    c->grid_first = c->grid = (p.npad + chunk - 1) / chunk;
    c->block = kWideBlock;
    return;
  }
  uint32_t block = env_u32("TSAMD_BLOCK", (c->cfg.k <= 16 && p.npairs >= 256u * 1024u) ? 512 : 256);
  if (block != 256u && block != 512u && block != 1024u) block = 256u;
  if (block == 1024u && c->cfg.k > 8) block = 512u;  // register budget of the pipelined loop
  auto geometry = [&](uint32_t nitems, uint32_t blk, uint32_t target, uint32_t &chunk, uint32_t &grid) {
    target = std::min<uint32_t>(std::max<uint32_t>(target, 1u), max_grid);
    chunk = (nitems + target - 1) / target;
    chunk = (chunk + blk - 1) / blk * blk;
    grid = (nitems + chunk - 1) / chunk;
  };
  c->block = block;
  c->first_vec = env_u32("TSAMD_FIRST_VEC", 1) == 2 ? 2 : 1;
  // Ranks that SHARE one device (tests, rehearsals: TSAMD_DEVICE_SHARE=<ranks>): a pass kernel of the peer-to-peer sequence
  // spins in its prologue until every rank's rows of the previous pass have arrived, so all ranks' kernels must fit the
  // device together -- a first pass that fills every compute unit (it is register-bound: one workgroup per unit from K = 12
  // on) would keep its peers' previous passes off the device until its bounded wait gives up (4 ranks x 250 000 individuals,
  // K = 20: "timed out waiting for a peer (epoch 2)").  Each rank gets its share of the workgroups.  One rank per device: 1.
  const uint32_t share = std::max<uint32_t>(1u, c->device_share);
  geometry(p.npairs, block, env_u32("TSAMD_GRID", share > 1u ? std::max<uint32_t>(8u, 256u / share) : 256u), p.chunk, c->grid);
  // first pass: exactly as many workgroups as are resident at once (one round; the kernel is
  // register-bound, so that is 2 per compute unit at K = 8 and 1 from K = 12 on)
  uint32_t first_target = 512;
  {
    hipDeviceProp_t prop;
    const int nb = kFirstBlocksPerCu[c->cfg.k]((int)c->first_vec);
    if (nb > 0 && hipGetDeviceProperties(&prop, c->dev) == hipSuccess && prop.multiProcessorCount > 0)
      first_target = (uint32_t)prop.multiProcessorCount * (uint32_t)std::min(nb, 4);
  }
  if (share > 1u) first_target = std::max<uint32_t>(8u, std::min<uint32_t>(first_target, 256u) / share);
  geometry(p.npad / c->first_vec, 256, env_u32("TSAMD_GRID_FIRST", first_target), p.chunk_first, c->grid_first);
}

// Launch geometry of the resident kernels for a shard of `npad` padded individuals on at most `cap` workgroups (all
// resident at once): items of resident_vec(K) individuals, a whole number of 256-thread rounds per workgroup.  False
// when the shard does not fit resident_items(K) items per thread.
// one_level: up to this many workgroups the kernel that will run exchanges in ONE level (ts_schedule: kResOneLevelGrid;
// ts_resident: 16 at K <= 8, never above -- its sweep's registers leave no room for the wider form).
bool resident_geometry(uint32_t k, uint32_t npad, uint32_t cap, uint32_t *grid, uint32_t *chunk, bool one_gpu = false,
                       uint32_t one_level = (uint32_t)kResOneLevelGrid) {
  if (cap == 0u || (int)k > kResidentMaxK) return false;
  const uint32_t nitems = npad / (uint32_t)resident_vec((int)k);
  auto rounds = [&](uint32_t workgroups) {
    const uint32_t ch = (nitems + workgroups - 1u) / workgroups;
    return (ch + (uint32_t)kResidentBlock - 1u) / (uint32_t)kResidentBlock;
  };
  uint32_t r = rounds(cap);
  // (a sharded launch -- one_gpu false -- holds sharded_items(K) items per thread: one fewer than resident_items(K) at K = 14 and 16)
  if (r > (uint32_t)(one_gpu ? resident_items((int)k) : sharded_items((int)k))) return false;
  // Small shards on one GPU: up to kResOneLevelGrid workgroups exchange in ONE level (1.9 us against 3.0 per pass), which is
  // worth a few more individuals per thread -- each costs about 0.33 K us per update (gamma step + ten sweeps), the nine
  // shorter exchanges save about 10 (profiles/r03_experiments.md)
  if (one_gpu && one_level > 0u && (nitems + r * (uint32_t)kResidentBlock - 1u) / (r * (uint32_t)kResidentBlock) > one_level) {
    const uint32_t r1 = rounds(one_level);
    if (r1 <= (uint32_t)resident_items((int)k) && (r1 - r) * k < 20u) r = r1;  // (measured with a threshold of 16: K = 8, N = 10 000: 39.0 against 43.7 us per update; K = 20, N = 8 000 would lose)
  }
  // ... and the smallest cohorts on ONE workgroup, which exchanges nothing at all (a pass is then a sweep, a fold and an
  // epilogue: about 1 us), when its extra individuals per thread cost less than the exchanges they replace
  if (one_gpu) {
    const uint32_t r0 = rounds(1u);
    if (r0 <= (uint32_t)resident_items((int)k) && (r0 - r) * k < 50u) r = r0;
  }
  *chunk = r * (uint32_t)kResidentBlock;
  *grid = (nitems + *chunk - 1u) / *chunk;
  return true;
}

// Launch geometry of ts_hybrid for a shard above ts_schedule's capacity: all `cap` workgroups, a whole number of 256-thread
// rounds each; the first hy_reg_items(K) + hy_lds_items(K) rounds of a workgroup stay on chip, the rest is streamed.
bool hybrid_geometry(uint32_t k, uint32_t npad, uint32_t cap, uint32_t *grid, uint32_t *chunk) {
  if (cap == 0u || (int)k > kResidentMaxK) return false;
  // (the kernel is bound by memory: ALL `cap` workgroups take an equal share -- a multiple of 16 individuals, i.e. of a column
  // word and of 128 bytes of a weight row -- rather than whole 256-thread rounds on fewer workgroups; a workgroup's last round
  // is then partly filled)
  const uint32_t ch = ((npad + cap - 1u) / cap + 15u) / 16u * 16u;
  const uint32_t r = (ch + (uint32_t)kResidentBlock - 1u) / (uint32_t)kResidentBlock;
  if (r > (uint32_t)(hy_reg_items((int)k) + hy_lds_items((int)k) + kHybridMaxStreamed)) return false;
  *chunk = ch;
  *grid = (npad + ch - 1u) / ch;
  return true;
}

bool alloc_res(tsamd_ctx *c) {
  if (c->res) return true;
  if (hipMalloc((void **)&c->res, sizeof(ResXchg)) != hipSuccess) return false;
  if (hipMemsetAsync(c->res, 0, sizeof(ResXchg), c->stream) != hipSuccess) return false;
  c->p.res = c->res;
  return true;
}

// ts_schedule on a shard: one launch per rank and schedule, weights resident, level 2 of the in-launch exchange across
// the ranks (Xchg::res_sums).  Every rank must reach the same verdict, so it depends only on the configuration: up to 8
// ranks, the reference's default learning-rate exponent, every rank's shard fits the register file of at most
// min(256, CUs / device_share) workgroups and fills at least 8 of them (all 8 groups of every rank then post a sum).
void choose_sharded_schedule(tsamd_ctx *c) {
  const tsamd_config &cfg = c->cfg;
  if (c->wide || cfg.world > 8u || (int)cfg.k > kResidentMaxK || cfg.nodekappa != 0.5 || cfg.max_inner < 2u || cfg.max_inner > 200u ||
      env_u32("TSAMD_RESIDENT", 1) == 0u || env_u32("TSAMD_PERSISTENT", 1) == 0u || kScheduleBlocksPerCu[cfg.k]() < 1)
    return;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, c->dev) != hipSuccess || prop.multiProcessorCount <= 0) return;
  // (ranks sharing a device: the dispatcher deals a launch's workgroups round robin over the 8 XCDs, every rank's launch starting at the
  // same one, so a rank may take floor(compute units per XCD / ranks) per XCD -- 3 ranks: 80 workgroups each, not 256 / 3 = 85, which put 33
  // workgroups on five XCDs of 32 compute units and lost the launch to the co-residency check: every 3-rank ts_hybrid test of round 5 in fact ran
  // its replay.  One rank per device: all compute units.)
  const uint32_t per_xcd = (uint32_t)prop.multiProcessorCount / (uint32_t)kResGroups;
  uint32_t cap = std::min<uint32_t>((uint32_t)(kResGroups * kResMembers),
                                    c->device_share > 1u ? (uint32_t)kResGroups * (per_xcd / c->device_share) : (uint32_t)prop.multiProcessorCount);
  if (env_u32("TSAMD_SCHED_WORKGROUPS", 0) >= (uint32_t)kResGroups) cap = std::min<uint32_t>(cap, env_u32("TSAMD_SCHED_WORKGROUPS", 0));  // (tuning knob, see tsamd_create)
  if (cap < (uint32_t)kResGroups) return;
  uint32_t my_grid = 0, my_chunk = 0;
  bool hybrid = false;
  for (int attempt = 0; attempt < 2; ++attempt) {
    // first ts_schedule on every rank; if a rank's shard exceeds its register capacity, ts_hybrid on every rank (up to 4 ranks:
    // the instantiations tsamd_hyb.hip carries)
    hybrid = attempt == 1;
    if (hybrid && (cfg.world > 4u || env_u32("TSAMD_HYBRID", 1) == 0u || kHybridBlocksPerCu[cfg.k]() < 1)) return;
    bool ok = true, too_big = false;
    for (uint32_t r = 0; r < cfg.world && ok; ++r) {
      uint32_t b = 0, cnt = 0, grid = 0, chunk = 0;
      tsamd_shard_range(cfg.n, r, cfg.world, &b, &cnt);
      const uint32_t npad_r = (cnt + 511u) / 512u * 512u;
      const bool fits = hybrid ? hybrid_geometry(cfg.k, npad_r, cap, &grid, &chunk) : resident_geometry(cfg.k, npad_r, cap, &grid, &chunk);
      too_big = too_big || !fits;
      ok = fits && grid >= (uint32_t)kResGroups;
      if (r == cfg.rank) {
        my_grid = grid;
        my_chunk = chunk;
      }
    }
    if (ok) break;
    // ts_hybrid is for shards ABOVE the register capacity.  A shard too SMALL to fill 8 workgroups (up to ~1 800 individuals:
    // not every group of every rank would post a sum) stays with one launch per pass and the peer-to-peer rows -- an
    // untuned hybrid geometry of 16 individuals per workgroup is not what such a run should get (advisor, round 4)
    if (hybrid || !too_big) return;
  }
  if (!alloc_res(c)) return;
  c->sched_grid = my_grid;
  c->sched_chunk = my_chunk;
  c->hybrid = hybrid;
  c->persistent = c->can_persistent = true;
  // (validation-mode schedules run batched on every rank alike: ts_holblock<K, WR>, level 2 of its wide exchange in Xchg::res_wide)
  c->can_holblock = !hybrid && kHolblockBlocksPerCu[cfg.k]() >= 1;
  c->can_hybhol = hybrid && kHybholBlocksPerCu[cfg.k]() >= 1;  // (... ts_hybhol<K, WR> when the ranks run ts_hybrid)
}

// Switch the kernel sequence to the exchange buffer (rows + epoch flags pushed by every
// workgroup to every rank); p.peers[] must be filled in.
void activate_xchg(tsamd_ctx *c) {
  c->p.xchg = c->xchg;
  c->p.xchg_world = c->cfg.world;
  c->p.xchg_rank = c->cfg.rank;
  c->p.rows_from_lt = 0u;
  // test hooks (tsamd_device.h): honoured only by a context created with TSAMD_FLAG_TEST_HOOKS -- a stray environment
  // variable must never switch the slot-reuse guard of a production run off
  const bool hooks = (c->cfg.flags & TSAMD_FLAG_TEST_HOOKS) != 0u;
  c->p.xchg_test_delay = hooks ? env_u32("TSAMD_TEST_XCHG_DELAY_US", 0) * 100u : 0u;
  c->p.xchg_test_noguard = hooks ? env_u32("TSAMD_TEST_XCHG_NOGUARD", 0) : 0u;
  {
    // ts_schedule across ranks: who gathers the ranks' group sums (all ranks must agree: the environment of a job)
    const char *g = getenv("TSAMD_SCHEDULE_GATHER");
    c->p.xchg_gather_leaders = (g && strcmp(g, "leaders") == 0) ? 1u : 0u;
  }
  c->split = true;
  c->resident = c->persistent = c->can_resident = c->can_persistent = c->hybrid = c->can_holblock = c->can_hybhol = false;
  c->p2p = true;
  configure_launch(c, std::max<uint32_t>(16u, std::min<uint32_t>(kXchgBlocks, 512u / c->cfg.world)));
  choose_sharded_schedule(c);
}

int alloc_xchg(tsamd_ctx *c) {
  if (c->xchg) return TSAMD_OK;
  HIP_TRY(c, hipExtMallocWithFlags((void **)&c->xchg, sizeof(Xchg), hipDeviceMallocFinegrained));
  HIP_TRY(c, hipMemset(c->xchg, 0, sizeof(Xchg)));
  return TSAMD_OK;
}

int ensure_stage(tsamd_ctx *c, size_t bytes) {
  if (c->stage_bytes >= bytes) return TSAMD_OK;
  if (c->h_stage) hipHostFree(c->h_stage);
  c->h_stage = nullptr;
  c->stage_bytes = 0;
  HIP_TRY(c, hipHostMalloc((void **)&c->h_stage, bytes, hipHostMallocDefault));
  c->stage_bytes = bytes;
  return TSAMD_OK;
}

// row-major [n_local][K] host -> k-major [K][npad] device (padding rows = pad)
int upload_kmajor(tsamd_ctx *c, const double *rows, double *dst, double pad) {
  const size_t K = c->cfg.k, np = c->npad, nl = c->n_local;
  std::vector<double> tmp(K * np, pad);
  for (size_t n = 0; n < nl; ++n)
    for (size_t k = 0; k < K; ++k) tmp[k * np + n] = rows[n * K + k];
  HIP_TRY(c, hipMemcpyAsync(dst, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return TSAMD_OK;
}

int export_indiv(tsamd_ctx *c, int mode, double *out) {
  const size_t K = c->cfg.k, nl = c->n_local;
  double *d_out = nullptr;
  HIP_TRY(c, hipMalloc((void **)&d_out, nl * K * sizeof(double)));
  hipLaunchKernelGGL(ts_export_indiv, dim3((nl + 255) / 256), dim3(256), 0, c->stream, c->p.gam, c->npad,
                     (uint32_t)K, (uint32_t)nl, (const uint32_t *)nullptr, mode, d_out);
  hipError_t e = hipMemcpyAsync(out, d_out, nl * K * sizeof(double), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  hipFree(d_out);
  if (e != hipSuccess) return fail(c, TSAMD_EHIP, "export_indiv: %s", hipGetErrorString(e));
  return TSAMD_OK;
}

int check_locs(tsamd_ctx *c, uint32_t first_loc, uint32_t n_locs) {
  if ((uint64_t)first_loc + n_locs > c->cfg.l)
    return fail(c, TSAMD_EINVAL, "location range [%u, %u) exceeds l = %u", first_loc, first_loc + n_locs, c->cfg.l);
  return TSAMD_OK;
}

}  // namespace

extern "C" {

// Wait for the stream, deal with what the kernels reported through the pinned error word (a resident launch that gave
// up at its entry is replayed one launch per pass), recycle the journal.  Defined with tsamd_run_schedule.
static int settle(tsamd_ctx *c);
// ... before anything that reads or replaces state while schedules may still be in flight
#define SETTLE(c)                          \
  do {                                     \
    if (!(c)->journal.empty())             \
      if (int rc_ = settle(c)) return rc_; \
  } while (0)

int tsamd_abi_version(void) { return TSAMD_ABI_VERSION; }

void tsamd_default_config(tsamd_config *cfg, uint32_t n, uint32_t l, uint32_t k) {
  memset(cfg, 0, sizeof *cfg);
  cfg->struct_size = sizeof *cfg;
  cfg->n = n;
  cfg->l = l;
  cfg->k = k;
  cfg->alpha = k ? 1.0 / (double)k : 0.0;
  cfg->eta0 = 1.0;
  cfg->eta1 = 1.0;
  cfg->nodetau0 = 2.0;
  cfg->nodekappa = 0.5;
  cfg->max_inner = 10;
  cfg->conv_thresh = 1e-3;
  cfg->gamma_scale = (double)l;
  cfg->device = 0;
  cfg->rank = 0;
  cfg->world = 1;
  cfg->flags = 0;
}

void tsamd_shard_range(uint32_t n, uint32_t rank, uint32_t world, uint32_t *begin, uint32_t *count) {
  if (world == 0) world = 1;
  uint64_t per = ((uint64_t)n + world - 1) / world;
  per = (per + 3) / 4 * 4;
  uint64_t b = std::min<uint64_t>((uint64_t)rank * per, n);
  uint64_t e = std::min<uint64_t>(b + per, n);
  if (begin) *begin = (uint32_t)b;
  if (count) *count = (uint32_t)(e - b);
}

const char *tsamd_last_error(const tsamd_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void tsamd_destroy(tsamd_ctx *c) {
  if (!c) return;
  hipSetDevice(c->dev);
  if (c->stream) hipStreamSynchronize(c->stream);
  destroy_graph(c);
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  for (void *m : c->peer_maps) hipIpcCloseMemHandle(m);
  if (c->xchg) hipFree(c->xchg);
  for (auto e : c->ev_pass) hipEventDestroy(e);
  for (auto e : c->ev_first) hipEventDestroy(e);
  hipFree(c->p.bed);
  hipFree(c->p.w);
  hipFree(c->p.gam);
  hipFree(c->p.cnt);
  hipFree(c->p.lam);
  hipFree(c->p.eb);
  hipFree(c->p.ctl);
  hipFree(c->p.partials);
  hipFree(c->d_sched);
  hipFree(c->res);
  if (c->h_error) hipHostFree(c->h_error);
  if (c->aux_stream) {
    hipStreamSynchronize(c->aux_stream);
    hipStreamDestroy(c->aux_stream);
  }
  if (c->h_occupy) hipHostFree(c->h_occupy);
  hipFree(c->d_hids);
  hipFree(c->d_hy);
  hipFree(c->d_hterms);
  hipFree(c->d_hreq);
  hipFree(c->d_fold_ids);
  hipFree(c->d_fold_orig);
  hipFree(c->d_hsums);
  if (c->h_stage) hipHostFree(c->h_stage);
  for (auto &j : c->journal) {
    hipHostFree(j.ent);
    if (j.done) hipEventDestroy(j.done);
  }
  for (auto e : c->event_free) hipEventDestroy(e);
  for (auto &b : c->sched_free) hipHostFree(b.first);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

int tsamd_create(const tsamd_config *cfg, tsamd_ctx **out) {
  if (!cfg || !out) return fail(nullptr, TSAMD_EINVAL, "null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(tsamd_config))
    return fail(nullptr, TSAMD_EINVAL, "tsamd_config size %u != %zu (ABI mismatch)", cfg->struct_size,
                sizeof(tsamd_config));
  if (cfg->n == 0 || cfg->l == 0 || cfg->k == 0) return fail(nullptr, TSAMD_EINVAL, "n, l, k must be positive");
  if (cfg->l >= 0x80000000u) return fail(nullptr, TSAMD_EINVAL, "l must be < 2^31");
  if (cfg->k > TSAMD_MAX_K)
    return fail(nullptr, TSAMD_EUNSUPPORTED, "k = %u above compiled maximum %d", cfg->k, TSAMD_MAX_K);
  if (cfg->max_inner == 0) return fail(nullptr, TSAMD_EINVAL, "max_inner must be >= 1");
  // (gamma never falls below min(gamma, alpha): with both >= 1e-8 -- the smallest non-zero value of the reference's %.8f
  // gamma.txt -- the exponent of exp(psi(gamma) - max) stays inside what exp_nonpos handles, |d| < 1.4e9)
  if (!(cfg->alpha >= 1e-8) || !std::isfinite(cfg->alpha)) return fail(nullptr, TSAMD_EINVAL, "alpha must be finite and >= 1e-8");
  if (!(cfg->eta0 > 0.0) || !(cfg->eta1 > 0.0) || !std::isfinite(cfg->eta0) || !std::isfinite(cfg->eta1))
    return fail(nullptr, TSAMD_EINVAL, "eta0, eta1 must be positive and finite");
  if (cfg->world == 0 || cfg->rank >= cfg->world) return fail(nullptr, TSAMD_EINVAL, "bad rank/world");
  uint32_t b = 0, cnt = 0;
  tsamd_shard_range(cfg->n, cfg->rank, cfg->world, &b, &cnt);
  if (cnt == 0) return fail(nullptr, TSAMD_EINVAL, "rank %u of %u owns no individuals (n = %u)", cfg->rank, cfg->world, cfg->n);

  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, TSAMD_EHIP, "no HIP device available (%s): libtsamd has no CPU path",
                hipGetErrorString(e));
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, TSAMD_EINVAL, "device %d out of range", cfg->device);

  tsamd_ctx *c = new tsamd_ctx();
  c->cfg = *cfg;
  // Ranks that share one device (tests, rehearsals) say so with TSAMD_DEVICE_SHARE=<ranks>: read ONCE, here, for sharded contexts
  // only -- every later decision (launch geometry, the sharded whole-schedule kernels, the geometry of a recovery) takes the share
  // from the context, so the ranks of a run cannot drift apart through their environments after they were created (advisor, round 5).
  // The ranks must be started with the same value; tsamd_p2p_connect_local sets it itself.
  if (cfg->world > 1u) c->device_share = std::max<uint32_t>(1u, env_u32("TSAMD_DEVICE_SHARE", 1));
  c->dev = cfg->device;
  c->n_begin = b;
  c->n_local = cnt;
  {
    // every rank pads to the same width, so that all ranks run the same launch geometry
    uint32_t b0 = 0, width = 0;
    tsamd_shard_range(cfg->n, 0, cfg->world, &b0, &width);
    c->npad = (std::max(width, cnt) + 511u) / 512u * 512u;
  }
#define CREATE_TRY(expr)                                                                          \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      int code_ = fail(nullptr, e_ == hipErrorOutOfMemory ? TSAMD_ENOMEM : TSAMD_EHIP, "%s: %s", \
                       #expr, hipGetErrorString(e_));                                             \
      tsamd_destroy(c);                                                                           \
      return code_;                                                                               \
    }                                                                                             \
  } while (0)
  CREATE_TRY(hipSetDevice(c->dev));
  CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));

  DevParams &p = c->p;
  const size_t K = cfg->k, np = c->npad, L = cfg->l;
  p.colstride = np / 4;  // npad multiple of 512 -> multiple of 128 bytes
  p.npad = c->npad;
  p.npairs = c->npad / 2;
  p.K = cfg->k;
  p.max_inner = cfg->max_inner;
  c->split = cfg->world > 1 || (cfg->flags & TSAMD_FLAG_SPLIT_EPILOGUE);
  p.alpha = cfg->alpha;
  p.eta0 = cfg->eta0;
  p.eta1 = cfg->eta1;
  p.nodetau0 = cfg->nodetau0;
  p.nodekappa = cfg->nodekappa;
  p.gamma_scale = cfg->gamma_scale;
  p.thresh = cfg->conv_thresh;

  c->wide = cfg->k > TSAMD_SPECIALIZED_K;
  configure_launch(c, kMaxGrid);
  if (c->wide && p.chunk_first > (uint32_t)kWideBlock * kWideItems) {
    fail(nullptr, TSAMD_EUNSUPPORTED, "k = %u (wide-K fallback) supports at most %u individuals per GPU", cfg->k,
         (unsigned)(kMaxGrid * kWideBlock * kWideItems));
    tsamd_destroy(c);
    return TSAMD_EUNSUPPORTED;
  }
  p.rows_from_lt = c->split ? 1u : 0u;
  p.sweep_alternate = env_u32("TSAMD_SWEEP", 1) ? 1u : 0u;
  p.probe_ticks = std::max<uint32_t>(1u, env_u32("TSAMD_PROBE_MS", 100)) * 100000u;  // (10 ns ticks)
  {
    // The resident kernels: one GPU (a sharded context decides in choose_sharded_schedule), K <= 32, the shard's
    // weights fit the register file (resident_items(K) items per thread of a 256-thread workgroup) and every workgroup
    // can be resident at once -- which the kernels verify for themselves at the start of every launch.
    hipDeviceProp_t prop;
    int cus = hipGetDeviceProperties(&prop, c->dev) == hipSuccess ? prop.multiProcessorCount : 0;
    // (test hook: fewer workgroups than the device holds, so that small shards exercise the many-items-per-thread paths --
    // ts_hybrid's LDS and streamed items for every K -- at a size the oracle finishes in a moment)
    if ((cfg->flags & TSAMD_FLAG_TEST_HOOKS) && env_u32("TSAMD_TEST_MAX_WORKGROUPS", 0) > 0u) cus = std::min<int>(cus, (int)env_u32("TSAMD_TEST_MAX_WORKGROUPS", 0));
    // (tuning knob, round 6's geometry sweep: at most this many workgroups for the resident kernels -- fewer members per exchange
    // group against more individuals per thread; profiles/r06_experiments.md.  Every rank of a sharded run must see the same value)
    if (env_u32("TSAMD_SCHED_WORKGROUPS", 0) >= (uint32_t)kResGroups) cus = std::min<int>(cus, (int)env_u32("TSAMD_SCHED_WORKGROUPS", 0));
    c->sched_grid = c->grid;
    c->sched_chunk = p.chunk;
    // (TSAMD_GRID / TSAMD_BLOCK shape the launch-per-pass kernels: a context they are set for runs those)
    const bool fits = !c->wide && cus > 0 && env_u32("TSAMD_GRID", 0) == 0u &&
                      resident_geometry(cfg->k, c->npad, std::min<uint32_t>((uint32_t)(kResGroups * kResMembers), (uint32_t)cus), &c->sched_grid,
                                        &c->sched_chunk, cfg->world == 1u);
    c->res_grid = c->sched_grid;
    c->res_chunk = c->sched_chunk;
    if (fits)  // (ts_resident<K> instantiates its exchange with one level up to 16 workgroups at K <= 8 and never above)
      resident_geometry(cfg->k, c->npad, std::min<uint32_t>((uint32_t)(kResGroups * kResMembers), (uint32_t)cus), &c->res_grid, &c->res_chunk,
                        cfg->world == 1u, cfg->k <= 8u ? 16u : 0u);
    c->resident = fits && !c->split && cfg->world == 1 && cfg->max_inner >= 2 && cfg->max_inner <= 200 &&
                  env_u32("TSAMD_RESIDENT", 1) != 0u && kResidentBlocksPerCu[cfg->k]() >= 1;
    // ... and then, with the reference's default learning-rate exponent (the kernel carries no pow()), the whole
    // schedule in one launch
    c->persistent = c->resident && cfg->nodekappa == 0.5 && env_u32("TSAMD_PERSISTENT", 1) != 0u &&
                    kScheduleBlocksPerCu[cfg->k]() >= 1;
    c->can_resident = c->resident;
    c->can_persistent = c->persistent;
    // A shard above that capacity: the same one-launch structure with part of the weights in LDS and the rest streamed
    // (ts_hybrid) instead of ten launches per update
    if (!fits && !c->wide && cus > 0 && !c->split && cfg->world == 1 && cfg->max_inner >= 2 && cfg->max_inner <= 200 && cfg->nodekappa == 0.5 &&
        env_u32("TSAMD_GRID", 0) == 0u && env_u32("TSAMD_RESIDENT", 1) != 0u && env_u32("TSAMD_PERSISTENT", 1) != 0u &&
        env_u32("TSAMD_HYBRID", 1) != 0u && kHybridBlocksPerCu[cfg->k]() >= 1 &&
        hybrid_geometry(cfg->k, c->npad, std::min<uint32_t>((uint32_t)(kResGroups * kResMembers), (uint32_t)cus), &c->sched_grid, &c->sched_chunk)) {
      c->hybrid = c->persistent = c->can_persistent = true;
    }
    c->can_holblock = c->can_persistent && !c->hybrid && cfg->world == 1u && kHolblockBlocksPerCu[cfg->k]() >= 1;  // (TSAMD_HOLBLOCK=0: read per call)
    c->can_hybhol = c->can_persistent && c->hybrid && cfg->world == 1u && kHybholBlocksPerCu[cfg->k]() >= 1;
  }

  CREATE_TRY(hipMalloc((void **)&p.bed, L * p.colstride));
  CREATE_TRY(hipMalloc((void **)&p.w, K * np * sizeof(double)));
  CREATE_TRY(hipMalloc((void **)&p.gam, K * np * sizeof(double)));
  CREATE_TRY(hipMalloc((void **)&p.cnt, np * sizeof(uint32_t)));
  CREATE_TRY(hipMalloc((void **)&p.lam, L * K * 2 * sizeof(double)));
  CREATE_TRY(hipMalloc((void **)&p.eb, L * K * 2 * sizeof(double)));
  CREATE_TRY(hipMalloc((void **)&p.ctl, sizeof(Ctl)));
  CREATE_TRY(hipMalloc((void **)&p.partials, (size_t)2 * kMaxGrid * 2 * TSAMD_MAX_K * sizeof(double)));
  c->sched_cap = 1024;
  CREATE_TRY(hipMalloc((void **)&c->d_sched, c->sched_cap * sizeof(uint32_t)));

  // pinned words the kernels write (DevParams::host_error): [0] error tag, [1] inner passes of the last completed SNP,
  // [2] total passes, [3 + b] pass histogram, [kHostDirtyWord] code of a wait that gave up after its launch had modified state
  CREATE_TRY(hipHostMalloc((void **)&c->h_error, (kHostDirtyWord + 1) * sizeof(unsigned long long), hipHostMallocDefault));
  memset(c->h_error, 0, (kHostDirtyWord + 1) * sizeof(unsigned long long));
  p.host_error = c->h_error;
  if ((c->resident || c->persistent) && !alloc_res(c)) CREATE_TRY(hipErrorOutOfMemory);
  CREATE_TRY(hipMemsetAsync(p.bed, 0x55, L * p.colstride, c->stream));  // all missing
  CREATE_TRY(hipMemsetAsync(p.cnt, 0, np * sizeof(uint32_t), c->stream));
  CREATE_TRY(hipMemsetAsync(p.ctl, 0, sizeof(Ctl), c->stream));
  CREATE_TRY(hipMemsetAsync(p.partials, 0, (size_t)2 * kMaxGrid * 2 * TSAMD_MAX_K * sizeof(double), c->stream));
  hipLaunchKernelGGL(ts_fill_f64, dim3(1024), dim3(256), 0, c->stream, p.gam, K * np, 1.0, 1.0);
  hipLaunchKernelGGL(ts_fill_f64, dim3(1024), dim3(256), 0, c->stream, p.w, K * np, 1.0, 1.0);
  // init_lambda (src/snpsamplinge.cc:239-250): lambda = eta, Elogbeta = psi(eta_t) - psi(eta0 + eta1)
  hipLaunchKernelGGL(ts_fill_f64, dim3(1024), dim3(256), 0, c->stream, p.lam, L * K * 2, cfg->eta0, cfg->eta1);
  {
    const uint64_t total = (uint64_t)L * K;
    const uint32_t per = 1u << 20;  // locations per launch
    for (uint64_t l0 = 0; l0 < L; l0 += per) {
      const uint32_t nl = (uint32_t)std::min<uint64_t>(per, L - l0);
      hipLaunchKernelGGL(ts_export_loc, dim3(((uint64_t)nl * K + 255) / 256), dim3(256), 0, c->stream, p.lam,
                         (uint32_t)K, (uint32_t)l0, nl, 2, p.eb);
    }
    (void)total;
  }
  CREATE_TRY(hipGetLastError());
  CREATE_TRY(hipStreamSynchronize(c->stream));
#undef CREATE_TRY
  *out = c;
  return TSAMD_OK;
}

// Shared by tsamd_upload_bed / _async.  Two routes:
//  * the payload lies in pinned host memory (tsamd_host_alloc, or registered by the caller): one
//    strided DMA straight from it (source pitch = bytes_per_snp, width = the shard's byte range,
//    destination pitch = the column stride) -- no staging copy, the padding bytes of every column
//    keep their "missing" fill from tsamd_create, a small kernel fixes the shard's last byte;
//  * pageable memory: through two pinned staging buffers, the host copy of one batch overlapping
//    the DMA of the previous one.
static int upload_bed_impl(tsamd_ctx *c, const uint8_t *payload, uint64_t bytes_per_snp, uint32_t first_loc,
                           uint32_t n_locs, bool wait) {
  CHECK_CTX(c);
  SETTLE(c);
  if (!payload) return fail(c, TSAMD_EINVAL, "null payload");
  if (bytes_per_snp != ((uint64_t)c->cfg.n + 3) / 4)
    return fail(c, TSAMD_EINVAL, "bytes_per_snp %llu != ceil(n/4) = %llu", (unsigned long long)bytes_per_snp,
                (unsigned long long)(((uint64_t)c->cfg.n + 3) / 4));
  if (int rc = check_locs(c, first_loc, n_locs)) return rc;
  if (n_locs == 0) return TSAMD_OK;
  HIP_TRY(c, hipSetDevice(c->dev));
  const size_t cs = c->p.colstride;
  const size_t src_off = c->n_begin / 4;
  const size_t nbytes = ((size_t)c->n_local + 3) / 4;
  const uint32_t tail = c->n_local & 3u;  // individuals in the last (partial) byte
  const uint8_t keep = (uint8_t)((1u << (2 * tail)) - 1u);
  hipPointerAttribute_t attr;
  const bool pinned = hipPointerGetAttributes(&attr, payload) == hipSuccess && attr.type == hipMemoryTypeHost;
  if (!pinned) (void)hipGetLastError();  // (an unknown pointer is reported as an error: pageable memory)
  if (pinned) {
    HIP_TRY(c, hipMemcpy2DAsync(c->p.bed + (size_t)first_loc * cs, cs, payload + src_off, bytes_per_snp, nbytes, n_locs,
                                hipMemcpyHostToDevice, c->stream));
    if (tail)
      hipLaunchKernelGGL(ts_fix_tail, dim3((n_locs + 255) / 256), dim3(256), 0, c->stream, c->p.bed, (uint64_t)cs, first_loc,
                         n_locs, (uint64_t)(nbytes - 1), (uint32_t)keep);
  } else {
    if (!wait) return fail(c, TSAMD_EINVAL, "tsamd_upload_bed_async needs pinned host memory (tsamd_host_alloc)");
    const size_t batch = std::max<size_t>(1, (size_t)(32u << 20) / cs);
    const size_t half = std::min<size_t>(batch, n_locs) * cs;
    if (int rc = ensure_stage(c, 2 * half)) return rc;
    hipEvent_t ev[2] = {nullptr, nullptr};
    HIP_TRY(c, hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    hipError_t e = hipSuccess;
    uint32_t b = 0;
    for (uint32_t j0 = 0; j0 < n_locs && e == hipSuccess; j0 += (uint32_t)batch, b ^= 1u) {
      const uint32_t nb = (uint32_t)std::min<size_t>(batch, n_locs - j0);
      uint8_t *stage = c->h_stage + (size_t)b * half;
      if (j0 >= 2 * batch) e = hipEventSynchronize(ev[b]);  // this half's previous DMA
      for (uint32_t j = 0; j < nb; ++j) {
        uint8_t *dst = stage + (size_t)j * cs;
        memcpy(dst, payload + (size_t)(j0 + j) * bytes_per_snp + src_off, nbytes);
        if (tail) dst[nbytes - 1] = (uint8_t)((dst[nbytes - 1] & keep) | (0x55u & ~keep));
        memset(dst + nbytes, 0x55, cs - nbytes);
      }
      if (e == hipSuccess)
        e = hipMemcpyAsync(c->p.bed + (size_t)(first_loc + j0) * cs, stage, (size_t)nb * cs, hipMemcpyHostToDevice, c->stream);
      if (e == hipSuccess) e = hipEventRecord(ev[b], c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipEventDestroy(ev[0]);
    hipEventDestroy(ev[1]);
    if (e != hipSuccess) return fail(c, TSAMD_EHIP, "upload_bed: %s", hipGetErrorString(e));
  }
  if (wait) HIP_TRY(c, hipStreamSynchronize(c->stream));
  // a re-upload drops validation folds of those columns
  for (auto it = c->held.lower_bound(first_loc); it != c->held.end() && it->first < first_loc + n_locs;)
    it = c->held.erase(it);
  c->held_dirty = true;
  return TSAMD_OK;
}

int tsamd_upload_bed(tsamd_ctx *c, const uint8_t *payload, uint64_t bytes_per_snp, uint32_t first_loc,
                     uint32_t n_locs) {
  return upload_bed_impl(c, payload, bytes_per_snp, first_loc, n_locs, true);
}

int tsamd_upload_bed_async(tsamd_ctx *c, const uint8_t *payload, uint64_t bytes_per_snp, uint32_t first_loc,
                           uint32_t n_locs) {
  return upload_bed_impl(c, payload, bytes_per_snp, first_loc, n_locs, false);
}

int tsamd_upload_bed_indiv_major(tsamd_ctx *c, const uint8_t *payload, uint64_t bytes_per_indiv, uint32_t first_indiv,
                                 uint32_t n_indivs) {
  CHECK_CTX(c);
  SETTLE(c);
  if (!payload) return fail(c, TSAMD_EINVAL, "null payload");
  if (bytes_per_indiv != ((uint64_t)c->cfg.l + 3) / 4)
    return fail(c, TSAMD_EINVAL, "bytes_per_indiv %llu != ceil(l/4) = %llu", (unsigned long long)bytes_per_indiv,
                (unsigned long long)(((uint64_t)c->cfg.l + 3) / 4));
  if ((uint64_t)first_indiv + n_indivs > c->cfg.n)
    return fail(c, TSAMD_EINVAL, "individuals [%u, %u) exceed n = %u", first_indiv, first_indiv + n_indivs, c->cfg.n);
  if (first_indiv % 16u != 0u) return fail(c, TSAMD_EINVAL, "first_indiv must be a multiple of 16 (whole column words)");
  // this shard's part of the batch
  const uint64_t lo = std::max<uint64_t>(first_indiv, c->n_begin), hi = std::min<uint64_t>((uint64_t)first_indiv + n_indivs,
                                                                                           (uint64_t)c->n_begin + c->n_local);
  if (lo >= hi) return TSAMD_OK;
  if ((lo - c->n_begin) % 16u != 0u)
    return fail(c, TSAMD_EUNSUPPORTED, "individual-major upload needs shard boundaries on multiples of 16 individuals");
  HIP_TRY(c, hipSetDevice(c->dev));
  const uint32_t rows_total = (uint32_t)(hi - lo), first_local = (uint32_t)(lo - c->n_begin);
  const size_t budget = (size_t)256u << 20;  // device staging: at most 256 MB of rows at a time
  const uint32_t per = (uint32_t)std::max<size_t>(kTrRows, std::min<size_t>(rows_total, budget / bytes_per_indiv / kTrRows * kTrRows));
  uint8_t *d_rows = nullptr;
  HIP_TRY(c, hipMalloc((void **)&d_rows, (size_t)per * bytes_per_indiv));
  hipError_t e = hipSuccess;
  for (uint32_t r0 = 0; r0 < rows_total && e == hipSuccess; r0 += per) {
    const uint32_t nr = std::min(per, rows_total - r0);
    e = hipMemcpyAsync(d_rows, payload + (size_t)(lo - first_indiv + r0) * bytes_per_indiv, (size_t)nr * bytes_per_indiv,
                       hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) break;
    const uint32_t nbytes = (uint32_t)bytes_per_indiv;
    for (uint32_t q0 = 0; q0 < nbytes; q0 += 65535u * kTrBytes) {  // (grid.y limit)
      const uint32_t nq = std::min<uint32_t>(nbytes - q0, 65535u * kTrBytes);
      const uint32_t locs0 = 4u * q0, nl = std::min<uint32_t>(c->cfg.l - locs0, 4u * nq);
      hipLaunchKernelGGL(ts_transpose_indiv_major, dim3((nr + kTrRows - 1) / kTrRows, (nq + kTrBytes - 1) / kTrBytes), dim3(256), 0,
                         c->stream, d_rows + q0, (uint64_t)bytes_per_indiv, nr, first_local + r0, locs0, nl, c->p.bed,
                         (uint64_t)c->p.colstride);
    }
    e = hipStreamSynchronize(c->stream);  // the staging buffer is reused
  }
  hipFree(d_rows);
  if (e != hipSuccess) return fail(c, TSAMD_EHIP, "upload_bed_indiv_major: %s", hipGetErrorString(e));
  c->held.clear();  // every column changed: validation folds are dropped
  c->held_dirty = true;
  return TSAMD_OK;
}

int tsamd_host_alloc(void **ptr, uint64_t bytes) {
  if (!ptr) return fail(nullptr, TSAMD_EINVAL, "null pointer");
  *ptr = nullptr;
  hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocPortable);
  if (e != hipSuccess) return fail(nullptr, e == hipErrorOutOfMemory ? TSAMD_ENOMEM : TSAMD_EHIP, "hipHostMalloc(%llu): %s",
                                   (unsigned long long)bytes, hipGetErrorString(e));
  return TSAMD_OK;
}

void tsamd_host_free(void *ptr) {
  if (ptr) (void)hipHostFree(ptr);
}

int tsamd_genotype_counts(tsamd_ctx *c, uint32_t first_loc, uint32_t n_locs, uint64_t counts[4]) {
  CHECK_CTX(c);
  SETTLE(c);
  if (!counts) return fail(c, TSAMD_EINVAL, "null output");
  if (int rc = check_locs(c, first_loc, n_locs)) return rc;
  for (int i = 0; i < 4; ++i) counts[i] = 0;
  if (n_locs == 0) return TSAMD_OK;
  HIP_TRY(c, hipSetDevice(c->dev));
  unsigned long long *d_out = (unsigned long long *)c->p.ctl->lt;  // scratch: 4 words of a buffer idle between schedules
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemsetAsync(d_out, 0, 4 * sizeof(unsigned long long), c->stream));
  const uint32_t nwords = (c->n_local + 31u) / 32u;
  const uint32_t gx = std::max<uint32_t>(1u, std::min<uint32_t>(64u, (nwords + 255u) / 256u));
  for (uint32_t j0 = 0; j0 < n_locs; j0 += 65535u) {
    const uint32_t nl = std::min<uint32_t>(65535u, n_locs - j0);
    hipLaunchKernelGGL(ts_count_codes, dim3(gx, nl), dim3(256), 0, c->stream, c->p.bed, (uint64_t)c->p.colstride,
                       first_loc + j0, c->n_local, d_out);
  }
  unsigned long long h[4] = {0, 0, 0, 0};
  HIP_TRY(c, hipMemcpyAsync(h, d_out, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemsetAsync(d_out, 0, 4 * sizeof(unsigned long long), c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < 4; ++i) counts[i] = h[i];
  return TSAMD_OK;
}

int tsamd_download_bed(tsamd_ctx *c, uint32_t loc, uint8_t *out, uint64_t out_bytes) {
  CHECK_CTX(c);
  SETTLE(c);
  if (int rc = check_locs(c, loc, 1)) return rc;
  const size_t nbytes = ((size_t)c->n_local + 3) / 4;
  if (!out || out_bytes < nbytes) return fail(c, TSAMD_EINVAL, "output buffer too small (%llu < %zu)",
                                             (unsigned long long)out_bytes, nbytes);
  HIP_TRY(c, hipSetDevice(c->dev));
  HIP_TRY(c, hipMemcpyAsync(out, c->p.bed + (size_t)loc * c->p.colstride, nbytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return TSAMD_OK;
}

int tsamd_set_heldout(tsamd_ctx *c, uint32_t loc, const uint32_t *indivs, uint32_t count) {
  CHECK_CTX(c);
  SETTLE(c);
  if (int rc = check_locs(c, loc, 1)) return rc;
  if (count && !indivs) return fail(c, TSAMD_EINVAL, "null indivs");
  HeldLoc &h = c->held[loc];
  c->held_dirty = true;
  // the new entries of this shard, ascending and without duplicates (the reference's map has one entry per
  // (individual, location); N/100 individuals per location at config 4: no quadratic searches)
  std::vector<uint32_t> ids;
  ids.reserve(count);
  for (uint32_t i = 0; i < count; ++i) {
    if (indivs[i] >= c->cfg.n) return fail(c, TSAMD_EINVAL, "individual %u >= n", indivs[i]);
    if (indivs[i] < c->n_begin || indivs[i] >= c->n_begin + c->n_local) continue;
    ids.push_back(indivs[i] - c->n_begin);
  }
  std::sort(ids.begin(), ids.end());
  ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
  if (!h.local_ids.empty())  // (kept ascending, see below)
    ids.erase(std::remove_if(ids.begin(), ids.end(),
                             [&](uint32_t lid) { return std::binary_search(h.local_ids.begin(), h.local_ids.end(), lid); }),
              ids.end());
  if (ids.empty()) {
    if (h.local_ids.empty()) c->held.erase(loc);
    return TSAMD_OK;
  }
  HIP_TRY(c, hipSetDevice(c->dev));
  // persistent scratch (ids in, original 2-bit codes out): no allocation per call
  if (ids.size() > c->fold_cap) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    hipFree(c->d_fold_ids);
    hipFree(c->d_fold_orig);
    c->d_fold_ids = nullptr;
    c->d_fold_orig = nullptr;
    c->fold_cap = 0;
    size_t cap = 4096;
    while (cap < ids.size()) cap *= 2;
    HIP_TRY(c, hipMalloc((void **)&c->d_fold_ids, cap * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->d_fold_orig, cap));
    c->fold_cap = cap;
  }
  uint32_t *d_ids = c->d_fold_ids;
  uint8_t *d_orig = c->d_fold_orig;
  std::vector<uint8_t> orig(ids.size());
  hipError_t e = hipMemcpyAsync(d_ids, ids.data(), ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(ts_heldout_fold, dim3((ids.size() + 255) / 256), dim3(256), 0, c->stream,
                       c->p.bed + (size_t)loc * c->p.colstride, d_ids, (uint32_t)ids.size(), d_orig);
    e = hipMemcpyAsync(orig.data(), d_orig, ids.size(), hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(c, TSAMD_EHIP, "set_heldout: %s", hipGetErrorString(e));
  static const uint8_t dec[4] = {0, 3, 1, 2};
  for (size_t i = 0; i < ids.size(); ++i) {
    if (dec[orig[i]] == 3) continue;  // was already missing: stays missing, not a held-out entry
    h.local_ids.push_back(ids[i]);
    h.ytrue.push_back(dec[orig[i]]);
  }
  // keep ascending individual order (compute_likelihood iterates the map in that order)
  std::vector<size_t> ord(h.local_ids.size());
  for (size_t i = 0; i < ord.size(); ++i) ord[i] = i;
  std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return h.local_ids[a] < h.local_ids[b]; });
  HeldLoc s;
  for (size_t i : ord) {
    s.local_ids.push_back(h.local_ids[i]);
    s.ytrue.push_back(h.ytrue[i]);
  }
  h = std::move(s);
  if (h.local_ids.empty()) c->held.erase(loc);
  return TSAMD_OK;
}

int tsamd_set_gamma(tsamd_ctx *c, const double *gamma) {
  CHECK_CTX(c);
  SETTLE(c);
  if (!gamma) return fail(c, TSAMD_EINVAL, "null gamma");
  for (size_t i = 0; i < (size_t)c->n_local * c->cfg.k; ++i)
    if (!(gamma[i] >= 1e-8) || !std::isfinite(gamma[i])) return fail(c, TSAMD_EINVAL, "gamma[%zu] must be finite and >= 1e-8", i);
  HIP_TRY(c, hipSetDevice(c->dev));
  if (int rc = upload_kmajor(c, gamma, c->p.gam, 1.0)) return rc;
  if (c->wide)
    hipLaunchKernelGGL(ts_refresh_w_wide, dim3((c->npad + 255) / 256), dim3(256), 0, c->stream, c->p);
  else
    kLaunchers[c->cfg.k](kLaunchRefresh, 0, 0, c->stream, c->p, 0, 0, 0u);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return TSAMD_OK;
}

int tsamd_get_gamma(tsamd_ctx *c, double *out) {
  CHECK_CTX(c);
  SETTLE(c);
  if (!out) return fail(c, TSAMD_EINVAL, "null output");
  HIP_TRY(c, hipSetDevice(c->dev));
  return export_indiv(c, 0, out);
}
int tsamd_get_theta(tsamd_ctx *c, double *out) {
  CHECK_CTX(c);
  SETTLE(c);
  if (!out) return fail(c, TSAMD_EINVAL, "null output");
  HIP_TRY(c, hipSetDevice(c->dev));
  return export_indiv(c, 1, out);
}
int tsamd_get_elogtheta(tsamd_ctx *c, double *out) {
  CHECK_CTX(c);
  SETTLE(c);
  if (!out) return fail(c, TSAMD_EINVAL, "null output");
  HIP_TRY(c, hipSetDevice(c->dev));
  return export_indiv(c, 2, out);
}

int tsamd_set_counts(tsamd_ctx *c, const uint32_t *cn) {
  CHECK_CTX(c);
  SETTLE(c);
  if (!cn) return fail(c, TSAMD_EINVAL, "null counts");
  HIP_TRY(c, hipSetDevice(c->dev));
  HIP_TRY(c, hipMemcpyAsync(c->p.cnt, cn, (size_t)c->n_local * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return TSAMD_OK;
}
int tsamd_get_counts(tsamd_ctx *c, uint32_t *cn) {
  CHECK_CTX(c);
  SETTLE(c);
  if (!cn) return fail(c, TSAMD_EINVAL, "null counts");
  HIP_TRY(c, hipSetDevice(c->dev));
  HIP_TRY(c, hipMemcpyAsync(cn, c->p.cnt, (size_t)c->n_local * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return TSAMD_OK;
}

int tsamd_set_lambda(tsamd_ctx *c, uint32_t loc, const double *lambda) {
  CHECK_CTX(c);
  SETTLE(c);
  if (!lambda) return fail(c, TSAMD_EINVAL, "null lambda");
  if (int rc = check_locs(c, loc, 1)) return rc;
  const size_t J = 2 * (size_t)c->cfg.k;
  for (size_t j = 0; j < J; ++j)
    if (!(lambda[j] > 0.0) || !std::isfinite(lambda[j])) return fail(c, TSAMD_EINVAL, "lambda must be positive and finite");
  HIP_TRY(c, hipSetDevice(c->dev));
  HIP_TRY(c, hipMemcpyAsync(c->p.lam + (size_t)loc * J, lambda, J * sizeof(double), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(ts_export_loc, dim3(1), dim3(256), 0, c->stream, c->p.lam, c->cfg.k, loc, 1u, 2, c->p.eb);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return TSAMD_OK;
}

static int export_loc(tsamd_ctx *c, uint32_t first_loc, uint32_t n_locs, int mode, double *out) {
  CHECK_CTX(c);
  SETTLE(c);
  if (!out) return fail(c, TSAMD_EINVAL, "null output");
  if (int rc = check_locs(c, first_loc, n_locs)) return rc;
  if (n_locs == 0) return TSAMD_OK;
  HIP_TRY(c, hipSetDevice(c->dev));
  const size_t K = c->cfg.k;
  const size_t per_loc = (mode == 0) ? K : 2 * K;
  if (mode < 0) {  // raw lambda
    HIP_TRY(c, hipMemcpyAsync(out, c->p.lam + (size_t)first_loc * 2 * K, (size_t)n_locs * 2 * K * sizeof(double),
                              hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return TSAMD_OK;
  }
  const uint32_t per = 1u << 18;
  double *d_out = nullptr;
  HIP_TRY(c, hipMalloc((void **)&d_out, (size_t)std::min(per, n_locs) * per_loc * sizeof(double)));
  hipError_t e = hipSuccess;
  for (uint32_t l0 = 0; l0 < n_locs && e == hipSuccess; l0 += per) {
    const uint32_t nl = std::min(per, n_locs - l0);
    hipLaunchKernelGGL(ts_export_loc, dim3(((uint64_t)nl * K + 255) / 256), dim3(256), 0, c->stream, c->p.lam,
                       (uint32_t)K, first_loc + l0, nl, mode, d_out);
    e = hipMemcpyAsync(out + (size_t)l0 * per_loc, d_out, (size_t)nl * per_loc * sizeof(double),
                       hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  hipFree(d_out);
  if (e != hipSuccess) return fail(c, TSAMD_EHIP, "export_loc: %s", hipGetErrorString(e));
  return TSAMD_OK;
}

int tsamd_get_lambda(tsamd_ctx *c, uint32_t first_loc, uint32_t n_locs, double *out) {
  return export_loc(c, first_loc, n_locs, -1, out);
}
int tsamd_get_ebeta(tsamd_ctx *c, uint32_t first_loc, uint32_t n_locs, double *out) {
  return export_loc(c, first_loc, n_locs, 0, out);
}
int tsamd_get_elogbeta(tsamd_ctx *c, uint32_t first_loc, uint32_t n_locs, double *out) {
  return export_loc(c, first_loc, n_locs, 1, out);
}

// Enqueue n schedule entries (location | hol << 31) that lie in pinned host memory, the way the context launches now.
// Everything that varies per SNP is read from device memory, so in the launch-per-pass mode captured sequences of
// 16, 8, 4, 2 and 1 SNPs are replayed as often as the schedule length needs (binary decomposition: nothing is padded);
// results are identical to eager launches bit for bit.  eager: no graphs (the replay after a failed resident launch).
static int enqueue_entries(tsamd_ctx *c, const uint32_t *ent, uint32_t n, bool eager, tsamd_ctx::Journal *jr = nullptr) {
  if (c->persistent) {
    // one launch runs the whole schedule (in pieces of kScheduleChunk SNPs): the kernel reads the entries straight from
    // the pinned buffer, one SNP ahead of their use; it starts from the State the previous call left and leaves one like
    // ts_flush does -- no ts_begin, no ts_flush, and none of the graphs of the launch-per-pass sequence
    auto launch = [&](bool hol_block, uint32_t off, uint32_t len) -> int {
      const bool prof = c->prof && c->n_ev_pass < kProfCap;
      if (c->prof && !prof) c->prof_capped = true;
      hipEvent_t e = nullptr;
      if (prof) {
        if (int rc = prof_event(c, c->ev_pass, 2 * c->n_ev_pass, &e)) return rc;
        HIP_TRY(c, hipEventRecord(e, c->stream));
      }
      if (jr) jr->launch_off.push_back(off);
      (hol_block ? (c->hybrid ? kHybholLaunchers : kHolblockLaunchers) : c->hybrid ? kHybridLaunchers : kScheduleLaunchers)[c->cfg.k](
          c->sched_grid, c->sched_chunk, c->stream, c->p, next_parity(c), ent + off, len, c->launch_serial++);
      if (hol_block) {
        c->holblock_launches++;
        c->holblock_locs += len;
      }
      if (prof) {
        if (int rc = prof_event(c, c->ev_pass, 2 * c->n_ev_pass + 1, &e)) return rc;
        HIP_TRY(c, hipEventRecord(e, c->stream));
        c->n_ev_pass++;
      }
      return TSAMD_OK;
    };
    uint32_t off = 0;
    // A validation-mode schedule (all entries of a call share the flag) leaves theta alone, so its locations are
    // independent as long as they are pairwise distinct: ts_holblock runs them in batches that share one sweep of the
    // weights per sub-batch and ONE exchange per pass.  Its first entry goes through ts_schedule when a training update
    // precedes it: that is where the pending gamma step is applied (src/snpsamplinge.cc:660-668).
    // (a context that runs ts_hybrid -- the shard exceeds the register capacity -- batches with ts_hybhol, which shares the
    // streamed weights across the locations of a sub-batch as well as the exchange; its first entry goes through ts_hybrid)
    if ((ent[0] >> 31) != 0u && (c->can_holblock || c->can_hybhol) && env_u32("TSAMD_HOLBLOCK", 1) != 0u && n >= (c->tail_step_pending ? 3u : 2u)) {
      if (c->tail_step_pending) {
        if (int rc = launch(false, 0u, 1u)) return rc;
        off = 1u;
      }
      std::unordered_set<uint32_t> seen;
      while (off < n) {
        seen.clear();
        uint32_t e = off;
        while (e < n && e - off < kHolChunk && seen.insert(ent[e] & 0x7fffffffu).second) ++e;
        if (int rc = launch(e - off >= 2u, off, e - off)) return rc;  // (a lone entry: ts_schedule does the same thing)
        off = e;
      }
    }
    for (; off < n; off += kScheduleChunk)
      if (int rc = launch(false, off, std::min(kScheduleChunk, n - off))) return rc;
    HIP_TRY(c, hipGetLastError());
    return TSAMD_OK;
  }
  if (n > c->sched_cap) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    hipFree(c->d_sched);
    c->d_sched = nullptr;
    c->sched_cap = 0;
    uint32_t cap = 1024;
    while (cap < n) cap *= 2;
    HIP_TRY(c, hipMalloc((void **)&c->d_sched, (size_t)cap * sizeof(uint32_t)));
    c->sched_cap = cap;  // (the kernels take the pointer from Ctl, written by ts_begin)
  }
  const bool use_graph = !eager && graphs_allowed(c);
  if (use_graph)
    if (int rc = ensure_graphs(c)) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->d_sched, ent, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  enqueue_begin(c, n, false);
  if (use_graph) {
    const uint32_t per_snp = kernels_per_snp(c);
    uint32_t left = n;
    auto replay = [&](uint32_t level) -> int {
      HIP_TRY(c, hipGraphLaunch(c->graphs[level][c->q & 1u].exec, c->stream));
      c->q += (uint64_t)(1u << level) * per_snp;
      left -= 1u << level;
      return TSAMD_OK;
    };
    // Submitting a 16-SNP graph (160 kernel nodes) costs the host ~90 us, during which an idle
    // device would wait, and every graph boundary costs the device ~8 us: so the first graph is a
    // small one (4 SNPs: the device starts after ~20 us and the larger submissions hide behind its
    // work), the rest are as few graphs as possible.  (Any order of graphs gives the same bits.)
    if (left > 4u)
      if (int rc = replay(2)) return rc;
    for (int level = (int)kGraphLevels - 1; level >= 0; --level)
      while (left >= (1u << level))
        if (int rc = replay((uint32_t)level)) return rc;
    c->prev_rows = c->cfg.max_inner > 1 ? c->grid : c->grid_first;
  } else {
    for (uint32_t i = 0; i < n; ++i)
      if (int rc = enqueue_snp(c)) return rc;
  }
  enqueue_flush(c);
  HIP_TRY(c, hipGetLastError());
  return TSAMD_OK;
}

// A resident launch gave up at its entry exchange -- not all its workgroups were resident at once: something else
// holds compute units of this device -- with the state it started from intact, and every later kernel of the context
// has returned without touching anything (sequence_aborted).  Lower the context to one launch per pass and replay,
// from the journal, what the failed launch and everything enqueued after it were to do.  The reference never loses
// the model to a scheduling hiccup either (it traps SIGTERM to save it, src/snpsamplinge.cc:454-457).
static int recover_from_failed_entry(tsamd_ctx *c, unsigned long long code) {
  // (the error word carries the low 30 bits of the launch serial: compare modulo 2^30)
  constexpr uint32_t kSerialMask = (1u << 30) - 1u;
  const uint32_t serial = (uint32_t)(code >> 34) & kSerialMask, par = (uint32_t)(code >> 33) & 1u;
  size_t at = c->journal.size();
  bool shrunk = false;  // ranks sharing a device: the replay (and everything after it) runs on a reduced launch geometry
  uint32_t ahead = 0;  // launches of the failed schedule before the failed one
  for (size_t i = 0; i < c->journal.size(); ++i) {
    const tsamd_ctx::Journal &j = c->journal[i];
    const uint32_t launches = j.mode == 2 ? (uint32_t)j.launch_off.size() : j.mode == 1 ? j.n : 0u;
    if (((serial - j.serial0) & kSerialMask) < launches) {
      at = i;
      ahead = (serial - j.serial0) & kSerialMask;
    }
  }
  if (at == c->journal.size())
    return fail(c, TSAMD_EHIP, "a resident launch (serial %u) gave up at its entry but is not in the journal of %zu schedule(s)", serial,
                c->journal.size());
  c->recovering = true;
  const bool was_persistent = c->journal[at].mode == 2;
  *(volatile unsigned long long *)c->h_error = 0ull;
  HIP_TRY(c, hipMemsetAsync(c->res, 0, sizeof(ResXchg), c->stream));  // the abort word, and the granules of the failed exchange
  if (c->p2p) {
    // the failed exchange's granules in the ranks' res_sums keep its tag, and peers may still be storing them: the next
    // resident launch (after tsamd_set_launch_mode raises the mode again) must not take them for its own -- skip the tags
    uint32_t xseq = 0;
    HIP_TRY(c, hipMemcpyAsync(&xseq, &c->p.ctl->xseq, sizeof xseq, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    xseq += 4096u;
    HIP_TRY(c, hipMemcpyAsync(&c->p.ctl->xseq, &xseq, sizeof xseq, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  c->resident = c->persistent = false;
  destroy_graph(c);
  if (c->p2p) {
    // Ranks that SHARE this device (tests, rehearsals): the resident launch failed because something holds compute units, and
    // the pass kernels of the replay spin in their prologues until every rank's previous pass has delivered its rows -- the
    // ranks' kernels must fit what is LEFT of the device together, or a rank's waiting workgroups keep its peers' previous
    // pass off it until the bounded waits give up ("the replay ... failed too": 3 ranks, K = 20 and a tenant on 200 of 256
    // compute units, one run in two).  The replay therefore runs on an eighth of each rank's share (every rank takes this
    // branch alike: the exchange's row layout stays consistent).  One rank per device: a pass kernel never waits for a
    // kernel that needs the same device, nothing to do.
    const uint32_t share = std::max<uint32_t>(1u, c->device_share);
    if (share > 1u) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, c->dev) != hipSuccess || prop.multiProcessorCount <= 0) {
        // (a rank that kept its grid while its peers shrank theirs would read mismatched rows: no silent skip)
        c->recovering = false;
        return fail(c, TSAMD_EHIP, "replay after a failed resident launch: cannot query device %d to size the replay's launches", c->dev);
      }
      configure_launch(c, std::max<uint32_t>(4u, (uint32_t)prop.multiProcessorCount / (8u * share)));
      shrunk = true;
    }
  }
  c->q = par;  // the failed launch was to write the slot of this parity: the slot of the other one holds the state to go on from
  int rc = TSAMD_OK;
  {
    const tsamd_ctx::Journal &j = c->journal[at];
    if (was_persistent) {
      const uint32_t off = j.launch_off[ahead];  // (first entry of the launch that gave up: ts_schedule's or ts_holblock's)
      rc = enqueue_entries(c, j.ent + off, j.n - off, true);
    } else {
      // ts_resident of SNP st.idx of this schedule: its first pass is done and pending (rows in the same slot); run its
      // plain passes, then the rest of the schedule
      State st;
      HIP_TRY(c, hipMemcpyAsync(&st, &c->p.ctl->st[par ^ 1u], sizeof(State), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      if (st.valid == 0u || st.done != 0u || st.idx >= j.n) {
        c->recovering = false;
        return fail(c, TSAMD_EHIP, "ts_resident gave up at its entry, but the state it started from has no pending pass (idx %u of %u)",
                    st.idx, j.n);
      }
      HIP_TRY(c, hipMemcpyAsync(c->d_sched, j.ent, (size_t)j.n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
      // The replayed kernels take the schedule from Ctl, which the ts_begin of THIS schedule set -- but a later schedule longer
      // than sched_cap may have been enqueued before the failure was noticed: enqueue_entries then freed and reallocated
      // d_sched while that schedule's own ts_begin was a no-op (sequence_aborted).  Point Ctl at the live buffer.
      {
        const uint32_t *live = c->d_sched;
        const uint32_t len = j.n;
        HIP_TRY(c, hipMemcpyAsync(&c->p.ctl->sched, &live, sizeof live, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(&c->p.ctl->sched_len, &len, sizeof len, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // (the sources are on this stack frame)
      }
      c->prev_rows = c->grid_first;
      for (uint32_t i = 1; rc == TSAMD_OK && i < c->cfg.max_inner; ++i) rc = enqueue_pass(c, i);
      for (uint32_t sn = st.idx + 1u; rc == TSAMD_OK && sn < j.n; ++sn) rc = enqueue_snp(c);
      if (rc == TSAMD_OK) enqueue_flush(c);
    }
  }
  for (size_t i = at + 1; rc == TSAMD_OK && i < c->journal.size(); ++i) rc = enqueue_entries(c, c->journal[i].ent, c->journal[i].n, true);
  if (rc == TSAMD_OK) {
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(c, TSAMD_EHIP, "replay after a failed resident launch: %s", hipGetErrorString(e));
  }
  c->recovering = false;
  if (rc != TSAMD_OK) return rc;
  if (*(volatile unsigned long long *)c->h_error != 0ull) return fail(c, TSAMD_EHIP, "the replay after a failed resident launch failed too");
  c->recoveries++;
  fail(c, TSAMD_OK, "warning: %s could not get its %u workgroups resident at once (something else holds compute units of device %d); the "
       "schedule was replayed one launch per pass from the unchanged state and the context stays in that mode "
       "(tsamd_set_launch_mode raises it again)%s", was_persistent ? (c->hybrid ? "ts_hybrid" : "ts_schedule") : "ts_resident", c->sched_grid, c->dev,
       shrunk ? "; ranks sharing the device: the launch-per-pass kernels keep the reduced geometry of the replay (an eighth of each rank's share) from here on" : "");
  return TSAMD_OK;
}

static int settle(tsamd_ctx *c) {
  HIP_TRY(c, hipSetDevice(c->dev));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  int rc = TSAMD_OK;
  if (c->h_error && *(volatile unsigned long long *)c->h_error != 0ull) {  // (written by the kernel that gave up)
    const unsigned long long err = *(volatile unsigned long long *)c->h_error;
    const unsigned long long tag = err & 0xffffffffull;
    // (a workgroup that passed the entry exchange, modified state and only then met the abort word leaves its code here:
    // the launch did NOT give up as a whole with its state intact, whatever the first word says)
    const bool dirty = *(volatile unsigned long long *)(c->h_error + kHostDirtyWord) != 0ull;
    if (getenv("TSAMD_DEBUG"))
      fprintf(stderr, "[tsamd rank %u] settle: error word %llx (tag %llu, intact %d, parity %d, serial %llu), modified-state word %llx, journal %zu, "
              "launch serial %u, mode %d/%d\n", c->cfg.rank, err, tag, (int)((err & kFailIntact) != 0ull), (int)((err >> 33) & 1ull), err >> 34,
              *(volatile unsigned long long *)(c->h_error + kHostDirtyWord), c->journal.size(), c->launch_serial, (int)c->resident, (int)c->persistent);
    // (a sharded context, one process per rank: the entry exchange spans the ranks, so it fails on EVERY rank -- nobody has
    // written anything -- and every rank, driven by the same calls, finds the same launch in its journal and replays the
    // same kernels.  Contexts of ONE process (tsamd_p2p_connect_local) are settled one after the other and would wait for
    // a peer's replay that has not been enqueued yet: they keep reporting TSAMD_ECOMM.)
    if (tag == 0xffffffffull)  // (tested first: on a sharded context the exchange branches below would report it as a peer that did not arrive)
      rc = fail(c, TSAMD_EHIP, "internal error: ts_holblock was launched with a gamma step pending (the state is intact; the context is not usable)");
    else if ((err & kFailIntact) != 0ull && !dirty && (c->cfg.world == 1u || (c->p2p && c->persistent && !c->peer_maps.empty())) && c->res && !c->recovering)
      rc = recover_from_failed_entry(c, err);
    else if (c->p2p && (c->persistent || (err & kFailIntact) != 0ull))
      rc = fail(c, TSAMD_ECOMM, "ts_schedule: the in-launch exchange across %u ranks timed out (tag %llu): a peer did not arrive, or "
                "not all workgroups of all ranks are resident (ranks that share one device: TSAMD_DEVICE_SHARE=<ranks>) [code %llx, "
                "modified-state word %llx, whole-schedule mode %d, mapped peers %zu]",
                c->cfg.world, tag, err, *(volatile unsigned long long *)(c->h_error + kHostDirtyWord), (int)c->persistent, c->peer_maps.size());
    else if (c->p2p)
      rc = fail(c, TSAMD_ECOMM, "peer-to-peer exchange timed out waiting for a peer (epoch %llu)", err);
    else
      rc = fail(c, TSAMD_EHIP, "%s: the in-launch exchange timed out in the middle of a launch (tag %llu, %u workgroups): the state is void.  "
                "TSAMD_PERSISTENT=0 selects one launch per SNP for the plain passes, TSAMD_RESIDENT=0 one launch per pass",
                c->persistent ? (c->hybrid ? "ts_hybrid" : "ts_schedule") : "ts_resident", tag, c->sched_grid);
  }
  for (auto &j : c->journal) {
    c->sched_free.push_back({j.ent, j.cap});
    if (j.done) c->event_free.push_back(j.done);
  }
  c->journal.clear();
  return rc;
}
int tsamd_run_schedule(tsamd_ctx *c, const uint32_t *locs, uint32_t n, int hol_mode) {
  CHECK_CTX(c);
  if (n == 0) return TSAMD_OK;
  if (!locs) return fail(c, TSAMD_EINVAL, "null schedule");
  if (c->cfg.world > 1 && !c->comm && !c->p2p)
    return fail(c, TSAMD_ECOMM, "context is shard %u of %u but neither tsamd_comm_init nor tsamd_p2p_connect has been called",
                c->cfg.rank, c->cfg.world);
  for (uint32_t i = 0; i < n; ++i)
    if (locs[i] >= c->cfg.l) return fail(c, TSAMD_EINVAL, "schedule[%u] = %u >= l", i, locs[i]);
  HIP_TRY(c, hipSetDevice(c->dev));
  // a caller that streams schedules and never synchronises must not grow the journal (pinned memory) without bound.
  // A peer-to-peer context cannot simply wait here (its kernels wait for peers the caller may not have enqueued yet): it
  // drops the entries whose kernels have finished -- an event per entry, queried, never waited for -- as long as no kernel
  // has reported anything (a failed launch is replayed from its own entry onwards: earlier ones are not needed)
  if (!c->p2p && c->journal.size() >= 256)
    if (int rc = settle(c)) return rc;
  if (c->p2p && c->journal.size() >= 256) {
    size_t drop = 0;
    while (drop < c->journal.size() && c->journal[drop].done && hipEventQuery(c->journal[drop].done) == hipSuccess &&
           *(volatile unsigned long long *)c->h_error == 0ull)
      ++drop;
    (void)hipGetLastError();  // (hipErrorNotReady of the first unfinished entry)
    for (size_t i = 0; i < drop; ++i) {
      c->sched_free.push_back({c->journal[i].ent, c->journal[i].cap});
      c->event_free.push_back(c->journal[i].done);
    }
    c->journal.erase(c->journal.begin(), c->journal.begin() + (long)drop);
    if (c->journal.size() >= 4096)  // (nothing finishes: the peers are not being driven -- the bounded waits will say so)
      if (int rc = settle(c)) return rc;
  }
  // A single update per call (tsamd_snp_update) through ts_schedule pays for loading the shard's weights and the LDS
  // half of gamma and writing them back around ONE update; from about 4M weights per GPU on, the launch-per-SNP
  // sequence is the faster route for such a call (N = 1M, K = 8: 7 830 against 6 860 calls/s; N = 125K, K = 20: 8 200
  // against 9 430, tools/single_update_rate.py).  The State either sequence leaves is the other's start.
  const bool single_route = n == 1u && c->persistent && c->can_resident && (uint64_t)c->n_local * c->cfg.k >= (4ull << 20) &&
                            env_u32("TSAMD_SINGLE_ROUTE", 1) != 0u;
  if (single_route) {
    c->persistent = false;
    c->resident = true;
  }
  // the schedule goes up through a pinned buffer: the copy is then really asynchronous
  tsamd_ctx::Journal j{nullptr, 0, n, c->launch_serial, c->persistent ? 2 : c->resident ? 1 : 0, {}, nullptr};
  for (size_t i = 0; i < c->sched_free.size(); ++i)
    if (c->sched_free[i].second >= n) {
      j.ent = c->sched_free[i].first;
      j.cap = c->sched_free[i].second;
      c->sched_free.erase(c->sched_free.begin() + i);
      break;
    }
  if (!j.ent) {
    size_t cap = 1024;
    while (cap < n) cap *= 2;
    // (portable + mapped: ts_schedule reads the entries straight from this buffer, on whichever device the context uses)
    if (hipHostMalloc((void **)&j.ent, cap * sizeof(uint32_t), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
      if (single_route) c->persistent = true;
      return fail(c, TSAMD_ENOMEM, "hipHostMalloc of a %zu-entry schedule buffer failed", cap);
    }
    j.cap = cap;
  }
  c->journal.push_back(j);
  for (uint32_t i = 0; i < n; ++i) j.ent[i] = locs[i] | (hol_mode ? 0x80000000u : 0u);
  int rc = enqueue_entries(c, j.ent, n, false, &c->journal.back());
  if (rc == TSAMD_OK) c->tail_step_pending = hol_mode == 0;  // (what was actually enqueued: a failed call leaves the flag as it was)
  if (rc == TSAMD_OK && c->p2p) {
    hipEvent_t ev = nullptr;
    if (!c->event_free.empty()) {
      ev = c->event_free.back();
      c->event_free.pop_back();
    } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      ev = nullptr;
      (void)hipGetLastError();
    }
    if (ev && hipEventRecord(ev, c->stream) == hipSuccess) c->journal.back().done = ev;
  }
  if (single_route) c->persistent = true;  // (c->resident stays set: it is what the mode falls back to when lowered by one)
  return rc;
}

int tsamd_prepare(tsamd_ctx *c) {
  CHECK_CTX(c);
  HIP_TRY(c, hipSetDevice(c->dev));
  if (c->cfg.world > 1 && !c->comm && !c->p2p) return TSAMD_OK;  // exchange not chosen yet: nothing to capture
  if (c->persistent) {  // an empty schedule: the kernel's code object is loaded, the state only carried forward
    (c->hybrid ? kHybridLaunchers : kScheduleLaunchers)[c->cfg.k](c->sched_grid, c->sched_chunk, c->stream, c->p, next_parity(c), c->d_sched, 0u,
                                                                  c->launch_serial++);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return TSAMD_OK;
  }
  if (!graphs_allowed(c)) return TSAMD_OK;
  if (int rc = ensure_graphs(c)) return rc;
  // One dry replay of every graph: whatever the runtime does on a graph's first launch happens
  // here.  With no schedule in progress every kernel of the sequence only carries the state
  // forward (sharded: all ranks call tsamd_prepare alike, so the launch sequences stay aligned).
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  hipLaunchKernelGGL(ts_begin, dim3(1), dim3(256), 0, c->stream, c->p, (const uint32_t *)nullptr, c->d_sched, 0u, next_parity(c), 0u);
  for (uint32_t level = 0; level < kGraphLevels; ++level)
    for (uint32_t par0 = 0; par0 < 2; ++par0) {
      if ((uint32_t)(c->q & 1u) != par0) enqueue_begin(c, 0xffffffffu, false);
      HIP_TRY(c, hipGraphLaunch(c->graphs[level][par0].exec, c->stream));
      c->q += (uint64_t)(1u << level) * kernels_per_snp(c);
    }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return TSAMD_OK;
}

int tsamd_synchronize(tsamd_ctx *c) {
  CHECK_CTX(c);
  if (int rc = settle(c)) return rc;
  if (c->prof) {
    for (uint32_t i = 0; i < c->n_ev_pass; ++i) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, c->ev_pass[2 * i], c->ev_pass[2 * i + 1]) == hipSuccess) {
        c->prof_pass_ms += ms;
        // launches inside the bracket (profile_read corrects for no-ops); ts_schedule: the bracket is one launch
        c->prof_pass_n += c->persistent ? 1u : c->cfg.max_inner - 1;
      }
    }
    for (uint32_t i = 0; i < c->n_ev_first; ++i) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, c->ev_first[2 * i], c->ev_first[2 * i + 1]) == hipSuccess) {
        c->prof_first_ms += ms;
        c->prof_first_n++;
      }
    }
    c->n_ev_pass = c->n_ev_first = 0;
  }
  return TSAMD_OK;
}

int tsamd_snp_update(tsamd_ctx *c, uint32_t loc, int hol_mode, uint32_t *inner_iters) {
  CHECK_CTX(c);
  if (int rc = tsamd_run_schedule(c, &loc, 1, hol_mode)) return rc;
  if (int rc = tsamd_synchronize(c)) return rc;
  if (inner_iters) {
    *inner_iters = (uint32_t) * (volatile unsigned long long *)(c->h_error + 1);  // (written by the kernel that completed the SNP)
  }
  return TSAMD_OK;
}

int tsamd_total_passes(tsamd_ctx *c, uint64_t *passes) {
  CHECK_CTX(c);
  if (!passes) return fail(c, TSAMD_EINVAL, "null output");
  if (int rc = settle(c)) return rc;
  *passes = *(volatile unsigned long long *)(c->h_error + 2);  // (mirrored by the kernel that publishes a SNP: no copy)
  return TSAMD_OK;
}

int tsamd_pass_histogram(tsamd_ctx *c, uint64_t hist[TSAMD_PASS_HIST_BINS]) {
  CHECK_CTX(c);
  if (!hist) return fail(c, TSAMD_EINVAL, "null output");
  if (int rc = settle(c)) return rc;
  static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "histogram element size");
  for (int b = 0; b < TSAMD_PASS_HIST_BINS; ++b) hist[b] = *(volatile unsigned long long *)(c->h_error + 3 + b);
  return TSAMD_OK;
}

int tsamd_clear_pending(tsamd_ctx *c) {
  CHECK_CTX(c);
  HIP_TRY(c, hipSetDevice(c->dev));
  SETTLE(c);
  enqueue_begin(c, 0xffffffffu, true);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->tail_step_pending = false;
  return TSAMD_OK;
}

// flat device table of the held-out entries (persistent; no allocation per evaluation)
static int sync_held_table(tsamd_ctx *c) {
  if (!c->held_dirty) return TSAMD_OK;
  size_t total = 0;
  for (auto &kv : c->held) total += kv.second.local_ids.size();
  std::vector<uint32_t> ids;
  std::vector<uint8_t> ys;
  ids.reserve(total);
  ys.reserve(total);
  c->held_span.clear();
  for (auto &kv : c->held) {
    if (kv.second.local_ids.empty()) continue;
    c->held_span[kv.first] = {ids.size(), (uint32_t)kv.second.local_ids.size()};
    ids.insert(ids.end(), kv.second.local_ids.begin(), kv.second.local_ids.end());
    ys.insert(ys.end(), kv.second.ytrue.begin(), kv.second.ytrue.end());
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (total > c->held_cap) {
    hipFree(c->d_hids);
    hipFree(c->d_hy);
    hipFree(c->d_hterms);
    c->d_hids = nullptr, c->d_hy = nullptr, c->d_hterms = nullptr, c->held_cap = 0;
    HIP_TRY(c, hipMalloc((void **)&c->d_hids, total * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->d_hy, total));
    HIP_TRY(c, hipMalloc((void **)&c->d_hterms, total * sizeof(double)));
    c->held_cap = total;
  }
  if (total) {
    HIP_TRY(c, hipMemcpy(c->d_hids, ids.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_hy, ys.data(), total, hipMemcpyHostToDevice));
  }
  c->held_dirty = false;
  return TSAMD_OK;
}

int tsamd_heldout_eval(tsamd_ctx *c, const uint32_t *locs, uint32_t n, int run_updates, double *loc_sums,
                       uint32_t *loc_counts, double *sum, uint32_t *count) {
  CHECK_CTX(c);
  if (sum) *sum = 0.0;
  if (count) *count = 0;
  if (n == 0) return TSAMD_OK;
  if (!locs) return fail(c, TSAMD_EINVAL, "null locations");
  for (uint32_t i = 0; i < n; ++i)
    if (int rc = check_locs(c, locs[i], 1)) return rc;
  HIP_TRY(c, hipSetDevice(c->dev));
  if (run_updates)
    if (int rc = tsamd_run_schedule(c, locs, n, 1)) return rc;
  if (int rc = tsamd_synchronize(c)) return rc;
  if (int rc = sync_held_table(c)) return rc;
  std::vector<HeldReq> req;       // requested locations that have held-out entries in this shard
  std::vector<uint32_t> req_of(n, 0xffffffffu);
  for (uint32_t i = 0; i < n; ++i) {
    auto it = c->held_span.find(locs[i]);
    if (it == c->held_span.end()) continue;
    req_of[i] = (uint32_t)req.size();
    req.push_back(HeldReq{it->second.first, it->second.second, locs[i]});
  }
  std::vector<double> sums(req.size(), 0.0);
  if (!req.empty()) {
    if (req.size() > c->hreq_cap) {
      hipFree(c->d_hreq);
      hipFree(c->d_hsums);
      c->d_hreq = nullptr, c->d_hsums = nullptr, c->hreq_cap = 0;
      size_t cap = 64;
      while (cap < req.size()) cap *= 2;
      HIP_TRY(c, hipMalloc((void **)&c->d_hreq, cap * sizeof(HeldReq)));
      HIP_TRY(c, hipMalloc((void **)&c->d_hsums, cap * sizeof(double)));
      c->hreq_cap = cap;
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_hreq, req.data(), req.size() * sizeof(HeldReq), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(ts_heldout_eval, dim3((uint32_t)req.size()), dim3(256), 0, c->stream, c->p.gam, c->npad, c->cfg.k,
                       c->p.lam, c->d_hids, c->d_hy, c->d_hreq, c->d_hterms, c->d_hsums);
    HIP_TRY(c, hipMemcpyAsync(sums.data(), c->d_hsums, req.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  double s = 0.0;
  uint32_t cnt = 0;
  for (uint32_t i = 0; i < n; ++i) {  // listed order, like compute_likelihood's loop over the map
    const bool has = req_of[i] != 0xffffffffu;
    const double u = has ? sums[req_of[i]] : 0.0;
    const uint32_t m = has ? req[req_of[i]].len : 0u;
    if (loc_sums) loc_sums[i] = u;
    if (loc_counts) loc_counts[i] = m;
    s += u;
    cnt += m;
  }
  if (sum) *sum = s;
  if (count) *count = cnt;
  return TSAMD_OK;
}

int tsamd_heldout_loglik(tsamd_ctx *c, uint32_t loc, double *sum, uint32_t *count) {
  return tsamd_heldout_eval(c, &loc, 1, 0, nullptr, nullptr, sum, count);
}

int tsamd_comm_unique_id(uint8_t id[TSAMD_COMM_ID_BYTES]) {
  if (!id) return fail(nullptr, TSAMD_EINVAL, "null id");
  static_assert(sizeof(ncclUniqueId) == TSAMD_COMM_ID_BYTES, "ncclUniqueId size");
  if (!g_rccl.load()) return fail(nullptr, TSAMD_ECOMM, "%s", g_rccl.error.c_str());
  ncclUniqueId uid;
  ncclResult_t r = g_rccl.GetUniqueId(&uid);
  if (r != ncclSuccess) return fail(nullptr, TSAMD_ECOMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
  memcpy(id, &uid, sizeof uid);
  return TSAMD_OK;
}

int tsamd_comm_init(tsamd_ctx *c, const uint8_t id[TSAMD_COMM_ID_BYTES]) {
  CHECK_CTX(c);
  if (!id) return fail(c, TSAMD_EINVAL, "null id");
  if (c->comm) return fail(c, TSAMD_EINVAL, "communicator already initialised");
  if (!g_rccl.load()) return fail(c, TSAMD_ECOMM, "%s", g_rccl.error.c_str());
  HIP_TRY(c, hipSetDevice(c->dev));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, (int)c->cfg.world, uid, (int)c->cfg.rank);
  if (r != ncclSuccess) {
    c->comm = nullptr;
    return fail(c, TSAMD_ECOMM, "ncclCommInitRank: %s", g_rccl.GetErrorString(r));
  }
  c->split = true;
  c->resident = c->persistent = c->can_resident = c->can_persistent = c->hybrid = c->can_holblock = c->can_hybhol = false;
  c->p.rows_from_lt = 1u;
  c->rccl_graph = env_u32("TSAMD_RCCL_GRAPH", 0) != 0u;
  destroy_graph(c);
  return TSAMD_OK;
}

int tsamd_p2p_export(tsamd_ctx *c, uint8_t handle[TSAMD_P2P_HANDLE_BYTES]) {
  CHECK_CTX(c);
  if (!handle) return fail(c, TSAMD_EINVAL, "null handle");
  static_assert(sizeof(hipIpcMemHandle_t) == TSAMD_P2P_HANDLE_BYTES, "hipIpcMemHandle_t size");
  if (c->cfg.world > (uint32_t)kMaxRanks) return fail(c, TSAMD_EUNSUPPORTED, "peer-to-peer exchange supports up to %d ranks", kMaxRanks);
  HIP_TRY(c, hipSetDevice(c->dev));
  if (int rc = alloc_xchg(c)) return rc;
  hipIpcMemHandle_t h;
  HIP_TRY(c, hipIpcGetMemHandle(&h, c->xchg));
  memcpy(handle, &h, sizeof h);
  return TSAMD_OK;
}

int tsamd_p2p_connect(tsamd_ctx *c, const uint8_t *handles) {
  CHECK_CTX(c);
  if (!handles) return fail(c, TSAMD_EINVAL, "null handles");
  if (!c->xchg) return fail(c, TSAMD_EINVAL, "tsamd_p2p_export has not been called");
  if (c->p2p) return fail(c, TSAMD_EINVAL, "peer-to-peer exchange already connected");
  HIP_TRY(c, hipSetDevice(c->dev));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (uint32_t q = 0; q < c->cfg.world; ++q) {
    if (q == c->cfg.rank) {
      c->p.peers[q] = c->xchg;
      continue;
    }
    hipIpcMemHandle_t h;
    memcpy(&h, handles + (size_t)q * TSAMD_P2P_HANDLE_BYTES, sizeof h);
    void *ptr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      for (void *m : c->peer_maps) hipIpcCloseMemHandle(m);
      c->peer_maps.clear();
      return fail(c, TSAMD_ECOMM, "hipIpcOpenMemHandle(rank %u): %s", q, hipGetErrorString(e));
    }
    c->peer_maps.push_back(ptr);
    c->p.peers[q] = (Xchg *)ptr;
  }
  activate_xchg(c);
  if (c->wide && c->p.chunk_first > (uint32_t)kWideBlock * kWideItems)
    return fail(c, TSAMD_EUNSUPPORTED, "wide-K fallback: shard too large for the peer-to-peer launch geometry");
  destroy_graph(c);
  return TSAMD_OK;
}

int tsamd_p2p_connect_local(tsamd_ctx *const *ctxs, uint32_t count) {
  if (!ctxs || count == 0) return fail(nullptr, TSAMD_EINVAL, "no contexts");
  tsamd_ctx *c0 = ctxs[0];
  CHECK_CTX(c0);
  if (count > (uint32_t)kMaxRanks) return fail(c0, TSAMD_EUNSUPPORTED, "peer-to-peer exchange supports up to %d ranks", kMaxRanks);
  std::vector<tsamd_ctx *> by_rank(count, nullptr);
  for (uint32_t i = 0; i < count; ++i) {
    tsamd_ctx *c = ctxs[i];
    if (!c) return fail(c0, TSAMD_EINVAL, "null context %u", i);
    if (c->cfg.world != count || c->cfg.rank >= count || by_rank[c->cfg.rank])
      return fail(c0, TSAMD_EINVAL, "contexts must be the ranks 0..%u of a world of %u, once each", count - 1, count);
    if (c->p2p || c->comm) return fail(c0, TSAMD_EINVAL, "context of rank %u already has an exchange", c->cfg.rank);
    if (c->cfg.n != c0->cfg.n || c->cfg.k != c0->cfg.k || c->cfg.l != c0->cfg.l)
      return fail(c0, TSAMD_EINVAL, "contexts differ in n / l / k");
    by_rank[c->cfg.rank] = c;
  }
  for (tsamd_ctx *c : by_rank) {
    HIP_TRY(c, hipSetDevice(c->dev));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (int rc = alloc_xchg(c)) return rc;
    for (tsamd_ctx *o : by_rank) {
      if (o->dev == c->dev) continue;
      int can = 0;
      HIP_TRY(c, hipDeviceCanAccessPeer(&can, c->dev, o->dev));
      if (!can) return fail(c, TSAMD_ECOMM, "device %d cannot access device %d", c->dev, o->dev);
      hipError_t e = hipDeviceEnablePeerAccess(o->dev, 0);
      if (e == hipErrorPeerAccessAlreadyEnabled)
        (void)hipGetLastError();
      else if (e != hipSuccess)
        return fail(c, TSAMD_ECOMM, "hipDeviceEnablePeerAccess(%d -> %d): %s", c->dev, o->dev, hipGetErrorString(e));
    }
  }
  uint32_t share = 1;  // contexts on the busiest device: their resident kernels must fit it together
  for (tsamd_ctx *c : by_rank) {
    uint32_t same = 0;
    for (tsamd_ctx *o : by_rank) same += o->dev == c->dev ? 1u : 0u;
    share = std::max(share, same);
  }
  for (tsamd_ctx *c : by_rank) {
    HIP_TRY(c, hipSetDevice(c->dev));
    for (uint32_t q = 0; q < count; ++q) c->p.peers[q] = by_rank[q]->xchg;
    c->device_share = share;
    activate_xchg(c);
    if (c->wide && c->p.chunk_first > (uint32_t)kWideBlock * kWideItems)
      return fail(c, TSAMD_EUNSUPPORTED, "wide-K fallback: shard too large for the peer-to-peer launch geometry");
    destroy_graph(c);
  }
  return TSAMD_OK;
}

int tsamd_run_schedule_all(tsamd_ctx *const *ctxs, uint32_t count, const uint32_t *locs, uint32_t n, int hol_mode) {
  if (!ctxs || count == 0 || !ctxs[0]) return fail(nullptr, TSAMD_EINVAL, "no contexts");
  // about 2K kernels per context and batch, a whole number of graph replays
  const uint32_t per_snp = std::max<uint32_t>(1u, ctxs[0]->cfg.max_inner);
  const uint32_t sub = std::max<uint32_t>(kGraphSnps, 2048u / per_snp / kGraphSnps * kGraphSnps);
  for (uint32_t off = 0; off < n; off += sub)
    for (uint32_t i = 0; i < count; ++i)
      if (int rc = tsamd_run_schedule(ctxs[i], locs ? locs + off : nullptr, std::min(sub, n - off), hol_mode)) return rc;
  return TSAMD_OK;
}

int tsamd_synth_genotypes(tsamd_ctx *c, const double *theta, const double *beta, uint32_t first_loc,
                          uint32_t n_locs, uint64_t seed, double missing_rate) {
  CHECK_CTX(c);
  SETTLE(c);
  if (!theta || !beta) return fail(c, TSAMD_EINVAL, "null theta/beta");
  if (int rc = check_locs(c, first_loc, n_locs)) return rc;
  if (n_locs == 0) return TSAMD_OK;
  HIP_TRY(c, hipSetDevice(c->dev));
  const size_t K = c->cfg.k, np = c->npad;
  double *d_theta = nullptr, *d_beta = nullptr;
  HIP_TRY(c, hipMalloc((void **)&d_theta, K * np * sizeof(double)));
  int rc = upload_kmajor(c, theta, d_theta, 0.0);
  const uint32_t per = 65535u * kSynthCols;
  if (rc == TSAMD_OK) {
    hipError_t e = hipMalloc((void **)&d_beta, (size_t)std::min(per, n_locs) * K * sizeof(double));
    for (uint32_t j0 = 0; j0 < n_locs && e == hipSuccess; j0 += per) {
      const uint32_t nl = std::min(per, n_locs - j0);
      e = hipMemcpyAsync(d_beta, beta + (size_t)j0 * K, (size_t)nl * K * sizeof(double), hipMemcpyHostToDevice, c->stream);
      if (e != hipSuccess) break;
      dim3 grid((np / 4 + kBlock - 1) / kBlock, (nl + kSynthCols - 1) / kSynthCols);
      hipLaunchKernelGGL(ts_synth, grid, dim3(kBlock), 0, c->stream, c->p.bed, c->p.colstride, d_theta, c->npad,
                         c->n_local, c->n_begin, (uint32_t)K, d_beta, first_loc + j0, nl, seed, missing_rate);
      e = hipStreamSynchronize(c->stream);
    }
    if (e != hipSuccess) rc = fail(c, TSAMD_EHIP, "synth_genotypes: %s", hipGetErrorString(e));
  }
  hipFree(d_theta);
  hipFree(d_beta);
  if (rc == TSAMD_OK)
    for (auto it = c->held.lower_bound(first_loc); it != c->held.end() && it->first < first_loc + n_locs;)
      it = c->held.erase(it);
  c->held_dirty = true;
  return rc;
}

int tsamd_profile_enable(tsamd_ctx *c, int on) {
  CHECK_CTX(c);
  if (int rc = tsamd_synchronize(c)) return rc;
  c->prof = on != 0;
  c->prof_pass_n = c->prof_first_n = 0;
  c->prof_pass_ms = c->prof_first_ms = 0;
  c->n_ev_pass = c->n_ev_first = 0;
  c->prof_capped = false;
  c->prof_passes0 = *(volatile unsigned long long *)(c->h_error + 2);  // (pinned mirror; the stream is idle: tsamd_synchronize above)
  return TSAMD_OK;
}

int tsamd_profile_read(tsamd_ctx *c, uint64_t *pass_launches, double *pass_ms_total, uint64_t *first_launches,
                       double *first_ms_total) {
  CHECK_CTX(c);
  if (int rc = tsamd_synchronize(c)) return rc;
  // plain passes that really swept: the passes the device counted since profiling was enabled
  // minus the first passes (a SNP that converges early leaves near-empty launches inside its
  // bracket; dividing by them would overstate the rate)
  if (!c->prof_capped && c->prof_first_n && !c->persistent) {
    const unsigned long long v = *(volatile unsigned long long *)(c->h_error + 2);
    const uint64_t ran = v - c->prof_passes0;
    if (ran >= c->prof_first_n && ran - c->prof_first_n <= c->prof_pass_n) c->prof_pass_n = ran - c->prof_first_n;
  }
  if (pass_launches) *pass_launches = c->prof_pass_n;
  if (pass_ms_total) *pass_ms_total = c->prof_pass_ms;
  if (first_launches) *first_launches = c->prof_first_n;
  if (first_ms_total) *first_ms_total = c->prof_first_ms;
  return TSAMD_OK;
}

int tsamd_probe_stream(tsamd_ctx *c, uint32_t reps, double *read_us, double *rmw_us) {
  CHECK_CTX(c);
  if (reps == 0) return fail(c, TSAMD_EINVAL, "reps must be positive");
  if (int rc = tsamd_synchronize(c)) return rc;
  HIP_TRY(c, hipSetDevice(c->dev));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIP_TRY(c, hipEventCreate(&e0));
  HIP_TRY(c, hipEventCreate(&e1));
  double *sink = c->p.partials;  // never written: the probe's condition cannot hold
  const uint32_t K = c->cfg.k;
  const uint32_t chunk_first = (!c->wide && c->first_vec == 2) ? c->p.chunk_first * 2u : c->p.chunk_first;  // individuals
  const uint32_t chunk_pairs = c->wide ? (c->p.chunk + 1u) / 2u : c->p.chunk;
  auto timed = [&](bool rmw, double *out_us) -> hipError_t {
    for (uint32_t r = 0; r < 3u + reps; ++r) {
      if (r == 3u) (void)hipEventRecord(e0, c->stream);
      if (rmw)
        hipLaunchKernelGGL(ts_probe_rmw, dim3(c->grid_first), dim3(256), 0, c->stream, c->p.w, c->p.gam, K, c->npad,
                           chunk_first, 1.0, env_u32("TSAMD_PROBE_THINK_NS", 0) / 10u);
      else
        hipLaunchKernelGGL(ts_probe_read, dim3(c->grid), dim3(c->block), 0, c->stream, c->p.w, K, c->npad, chunk_pairs,
                           c->p.sweep_alternate ? (r & 1u) : 0u, sink);
    }
    hipError_t e = hipEventRecord(e1, c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (out_us) *out_us = (double)ms * 1e3 / reps;
    return e;
  };
  hipError_t e = timed(false, read_us);
  if (e == hipSuccess) e = timed(true, rmw_us);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  if (e != hipSuccess) return fail(c, TSAMD_EHIP, "probe_stream: %s", hipGetErrorString(e));
  return TSAMD_OK;
}

int tsamd_launch_info(tsamd_ctx *c, uint32_t *kernels_per_snp_out, uint32_t *plain_grid, uint32_t *first_grid) {
  CHECK_CTX(c);
  if (kernels_per_snp_out) *kernels_per_snp_out = c->persistent ? 0u : kernels_per_snp(c);
  if (plain_grid) *plain_grid = c->grid;
  if (first_grid) *first_grid = c->grid_first;
  return TSAMD_OK;
}

int tsamd_schedule_geometry(tsamd_ctx *c, int mode, uint32_t *workgroups, uint32_t *indivs_per_thread, uint32_t *exchange_levels,
                            uint32_t *on_chip_per_thread) {
  CHECK_CTX(c);
  if (mode != TSAMD_LAUNCH_PER_SNP && mode != TSAMD_LAUNCH_PER_SCHEDULE) return fail(c, TSAMD_EINVAL, "launch mode %d has no resident kernel", mode);
  if ((mode == TSAMD_LAUNCH_PER_SNP && !c->can_resident) || (mode == TSAMD_LAUNCH_PER_SCHEDULE && !c->can_persistent))
    return fail(c, TSAMD_EUNSUPPORTED, "the context does not qualify for launch mode %d", mode);
  const bool sched = mode == TSAMD_LAUNCH_PER_SCHEDULE;
  const uint32_t grid = sched ? c->sched_grid : c->res_grid, chunk = sched ? c->sched_chunk : c->res_chunk;
  const uint32_t one = sched ? (uint32_t)kResOneLevelGrid : (c->cfg.k <= 8u ? 16u : 0u);
  if (workgroups) *workgroups = grid;
  if (indivs_per_thread) *indivs_per_thread = (chunk + (uint32_t)kResidentBlock - 1u) / (uint32_t)kResidentBlock * (uint32_t)resident_vec((int)c->cfg.k);
  if (exchange_levels) *exchange_levels = (grid == 1u && c->cfg.world == 1u) ? 0u : (c->cfg.world == 1u && grid <= one) ? 1u : 2u;
  if (on_chip_per_thread) {
    const uint32_t per = (chunk + (uint32_t)kResidentBlock - 1u) / (uint32_t)kResidentBlock * (uint32_t)resident_vec((int)c->cfg.k);
    *on_chip_per_thread = (sched && c->hybrid) ? std::min<uint32_t>(per, (uint32_t)(hy_reg_items((int)c->cfg.k) + hy_lds_items((int)c->cfg.k))) : per;
  }
  return TSAMD_OK;
}

int tsamd_holblock_info(tsamd_ctx *c, uint32_t *batch, uint64_t *launches, uint64_t *locations) {
  CHECK_CTX(c);
  if (batch)
    *batch = (c->persistent && env_u32("TSAMD_HOLBLOCK", 1) != 0u) ? (c->can_holblock ? (uint32_t)hol_batch((int)c->cfg.k) : c->can_hybhol ? (uint32_t)kHybholBatch[c->cfg.k]() : 0u) : 0u;
  if (launches) *launches = c->holblock_launches;
  if (locations) *locations = c->holblock_locs;
  return TSAMD_OK;
}

int tsamd_set_launch_mode(tsamd_ctx *c, int mode) {
  CHECK_CTX(c);
  if (mode < TSAMD_LAUNCH_PER_PASS || mode > TSAMD_LAUNCH_PER_SCHEDULE) return fail(c, TSAMD_EINVAL, "launch mode %d", mode);
  if ((mode == TSAMD_LAUNCH_PER_SNP && !c->can_resident) || (mode == TSAMD_LAUNCH_PER_SCHEDULE && !c->can_persistent))
    return fail(c, TSAMD_EUNSUPPORTED, "launch mode %d needs k <= %d, a shard that fits the register file (%d individuals per workgroup at k = %u)%s",
                mode, kResidentMaxK, (int)c->cfg.k <= kResidentMaxK ? resident_capacity((int)c->cfg.k) : 0, c->cfg.k,
                mode == TSAMD_LAUNCH_PER_SCHEDULE ? " and nodekappa == 0.5" : " and one GPU");
  if (int rc = tsamd_synchronize(c)) return rc;
  const bool resident = mode >= TSAMD_LAUNCH_PER_SNP && c->can_resident, persistent = mode == TSAMD_LAUNCH_PER_SCHEDULE;
  if (resident != c->resident) destroy_graph(c);  // (captured for the other kernel sequence)
  c->resident = resident;  // (a sharded context has no launch-per-SNP mode: ts_schedule or one launch per pass)
  c->persistent = persistent;
  return TSAMD_OK;
}

int tsamd_debug_occupy(tsamd_ctx *c, uint32_t workgroups, uint32_t milliseconds) {
  CHECK_CTX(c);
  if (workgroups == 0 || milliseconds > 10000u) return fail(c, TSAMD_EINVAL, "workgroups must be positive, milliseconds <= 10000");
  HIP_TRY(c, hipSetDevice(c->dev));
  if (!c->aux_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
  if (!c->h_occupy) HIP_TRY(c, hipHostMalloc((void **)&c->h_occupy, sizeof(unsigned long long), hipHostMallocDefault));
  HIP_TRY(c, hipStreamSynchronize(c->aux_stream));
  *(volatile unsigned long long *)c->h_occupy = 0ull;
  hipLaunchKernelGGL(ts_occupy, dim3(workgroups), dim3(256), 0, c->aux_stream, (unsigned long long)milliseconds * 100000ull, c->h_occupy);
  HIP_TRY(c, hipGetLastError());
  for (int spin = 0; spin < 2000000 && *(volatile unsigned long long *)c->h_occupy == 0ull; ++spin) {  // until it runs (bounded: ~2 s)
    if (hipStreamQuery(c->aux_stream) == hipSuccess) break;
  }
  return TSAMD_OK;
}

int tsamd_recoveries(tsamd_ctx *c, uint32_t *count) {
  CHECK_CTX(c);
  if (!count) return fail(c, TSAMD_EINVAL, "null output");
  *count = c->recoveries;
  return TSAMD_OK;
}

int tsamd_mem_info(tsamd_ctx *c, uint64_t *free_bytes, uint64_t *total_bytes) {
  CHECK_CTX(c);
  HIP_TRY(c, hipSetDevice(c->dev));
  size_t f = 0, t = 0;
  HIP_TRY(c, hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return TSAMD_OK;
}

}  // extern "C"
