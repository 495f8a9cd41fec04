/*
 * tsamd.h -- C ABI of the MI355X-native SNP-minibatch SVI engine
 * (libtsamd.so, built from terastructure_amd/csrc/ with hipcc for gfx950).
 *
 * The reference has no plugin/FFI layer: its C++ host calls
 * SNPSamplingE::optimize_lambda(loc) directly from the main thread
 * (callers: src/snpsamplinge.cc:425 infer, :374 compute_all_lambda,
 * :405 compute_and_save_beta, src/snpsamplinge.hh:332 snp_likelihood).
 * This header is that seam as a C ABI: plain pointers and sizes, fp64
 * row-major arrays in the reference's shapes, no torch/HIP types.
 *
 * Conventions
 *  - every call returns 0 on success or a negative TSAMD_E* code and never
 *    calls exit()/assert() (the reference does, e.g. src/snpsamplinge.cc:69-72);
 *    tsamd_last_error() gives the message.
 *  - one caller thread per context; calls are synchronous unless stated.
 *  - a context owns one shard of individuals [shard_begin, shard_begin +
 *    shard_count) given by tsamd_shard_range(n, rank, world); per-individual
 *    arrays passed in/out cover THAT shard only (world == 1: everyone).
 *  - locations are 0-based SNP indices < l.
 */
#ifndef TSAMD_H
#define TSAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSAMD_ABI_VERSION 1

#define TSAMD_OK 0
#define TSAMD_EINVAL (-1)       /* bad argument */
#define TSAMD_EHIP (-2)         /* HIP runtime error */
#define TSAMD_ENOMEM (-3)       /* host or device allocation failed */
#define TSAMD_ECOMM (-4)        /* RCCL error / communicator missing */
#define TSAMD_EUNSUPPORTED (-5) /* e.g. K above the compiled maximum */

#define TSAMD_MAX_K 128          /* populations; K <= TSAMD_SPECIALIZED_K runs the K-specialised kernels, */
#define TSAMD_SPECIALIZED_K 32  /* larger K a slower run-time-K fallback */
#define TSAMD_COMM_ID_BYTES 128

typedef struct tsamd_ctx tsamd_ctx;

/* Model constants; defaults are the reference's compile-time values. */
typedef struct tsamd_config {
  uint32_t struct_size; /* sizeof(tsamd_config), ABI check */
  uint32_t n;           /* individuals, global            (-n, src/main.cc:115) */
  uint32_t l;           /* SNP locations                  (-l, src/main.cc:121) */
  uint32_t k;           /* populations                    (-k, src/main.cc:118) */
  double alpha;         /* 1/k                            src/env.hh:209 */
  double eta0, eta1;    /* 1, 1                           src/env.hh:221-222 */
  double nodetau0;      /* 2 (env 1, +1 in the engine)    src/snpsamplinge.cc:16 */
  double nodekappa;     /* 0.5                            src/env.hh:230 */
  uint32_t max_inner;   /* online_iterations: 10; 100 in -compute-beta  src/env.hh:232, src/snpsamplinge.cc:75 */
  double conv_thresh;   /* meanchangethresh 1e-3          src/env.hh:208 */
  double gamma_scale;   /* = l                            src/snpsamplinge.cc:702 */
  int32_t device;       /* HIP device ordinal */
  uint32_t rank, world; /* individual sharding; world == 1 for a single GPU */
  uint32_t flags;       /* TSAMD_FLAG_* */
} tsamd_config;

#define TSAMD_FLAG_SPLIT_EPILOGUE 1u /* run the sharded kernel sequence (pass, row sum, exchange) even on one GPU */
#define TSAMD_FLAG_NO_GRAPH 2u       /* tsamd_run_schedule launches eagerly instead of replaying a hipGraph */
#define TSAMD_FLAG_TEST_HOOKS 4u     /* honour the TSAMD_TEST_* environment hooks (tests only): those of the peer-to-peer exchange, and
                                        TSAMD_TEST_MAX_WORKGROUPS (the resident kernels' launch geometry as on a device with that
                                        many compute units: small shards then run the many-items-per-thread paths) */

int tsamd_abi_version(void);
void tsamd_default_config(tsamd_config *cfg, uint32_t n, uint32_t l, uint32_t k);
/* contiguous shards of ceil(n/world) rounded up to a multiple of 4 individuals
 * (byte-aligned slices of a .bed column); mirrors split_all_indivs
 * (src/snpsamplinge.cc:298-318) with GPUs in place of threads. */
void tsamd_shard_range(uint32_t n, uint32_t rank, uint32_t world, uint32_t *begin, uint32_t *count);

/* replaces SNPSamplingE::SNPSamplingE state allocation + init_lambda
 * (src/snpsamplinge.cc:6-37, :239-250): lambda = eta, gamma = 1, c_n = 0,
 * all genotypes missing until uploaded. */
int tsamd_create(const tsamd_config *cfg, tsamd_ctx **out);
void tsamd_destroy(tsamd_ctx *ctx);
/* message of the last failing call on ctx (ctx == NULL: last tsamd_create failure) */
const char *tsamd_last_error(const tsamd_ctx *ctx);

/* replaces SNP::read_bed's decode loop (src/snp.cc:195-228): raw PLINK
 * SNP-major payload after the 3 magic bytes, n_locs columns of bytes_per_snp =
 * ceil(n/4) bytes (GLOBAL n); the context copies its shard's byte range and
 * keeps the 2-bit codes packed in HBM.  The host buffer is not retained. */
int tsamd_upload_bed(tsamd_ctx *ctx, const uint8_t *payload, uint64_t bytes_per_snp,
                     uint32_t first_loc, uint32_t n_locs);
/* Streaming ingest.  tsamd_host_alloc gives pinned host memory; a payload that lies
 * in it is copied by ONE strided DMA per call straight from the caller's buffer (no staging copy
 * on the host), whereas pageable memory goes through the library's two pinned staging buffers.
 * tsamd_upload_bed_async (pinned payloads only) returns after enqueueing: the buffer must stay
 * unchanged until the next tsamd_synchronize(ctx) -- read the next batch of the file into a second
 * buffer meanwhile.  Several contexts (shards) may upload from the same buffer. */
int tsamd_host_alloc(void **ptr, uint64_t bytes);
void tsamd_host_free(void *ptr);
int tsamd_upload_bed_async(tsamd_ctx *ctx, const uint8_t *payload, uint64_t bytes_per_snp,
                           uint32_t first_loc, uint32_t n_locs);
/* PLINK's other .bed layout, individual-major (third magic byte 0), which the reference refuses
 * ("individual major mode not supported yet!", src/snp.cc:176-178): n_indivs rows of
 * bytes_per_indiv = ceil(l/4) bytes, row r = GLOBAL individual first_indiv + r, location j at bits
 * 2(j%4) of byte j/4, same 2-bit codes.  The rows are transposed on the device into the SNP-major
 * columns the engine keeps.  first_indiv must be a multiple of 16; a batch may be any number of
 * rows (the last word's missing individuals stay "missing" until a later batch brings them, so
 * upload in ascending order).  A context takes the rows of its own shard.  Drops all held-out folds. */
int tsamd_upload_bed_indiv_major(tsamd_ctx *ctx, const uint8_t *payload, uint64_t bytes_per_indiv,
                                 uint32_t first_indiv, uint32_t n_indivs);
/* replaces the genotype tallies of SNP::read_bed (src/snp.cc:203-216, reported in param.txt,
 * :245-247): counts[c] = entries with PLINK code c (00, 01 = missing, 10, 11) among the shard's
 * individuals in columns [first_loc, first_loc + n_locs), counted on the device.  Held-out
 * entries count as missing: call it before tsamd_set_heldout for the reference's numbers. */
int tsamd_genotype_counts(tsamd_ctx *ctx, uint32_t first_loc, uint32_t n_locs, uint64_t counts[4]);
/* shard's column as ceil(shard_count/4) PLINK bytes, held-out entries shown as missing */
int tsamd_download_bed(tsamd_ctx *ctx, uint32_t loc, uint8_t *out, uint64_t out_bytes);

/* replaces _validation_map inserts (src/snpsamplinge.cc:213-217) and their
 * kv_ok effect (src/snpsamplinge.hh:389-408): (indiv, loc) then behaves as
 * missing in every pass and gamma step.  indivs are GLOBAL ids; ids outside
 * the shard are ignored.  True genotypes are kept for tsamd_heldout_loglik. */
int tsamd_set_heldout(tsamd_ctx *ctx, uint32_t loc, const uint32_t *indivs, uint32_t count);

/* gamma [shard_count][k] row-major.  set also refreshes Elogtheta
 * (estimate_all_theta, src/snpsamplinge.cc:595-609); c_n is left alone. */
int tsamd_set_gamma(tsamd_ctx *ctx, const double *gamma);
int tsamd_get_gamma(tsamd_ctx *ctx, double *gamma);
int tsamd_get_theta(tsamd_ctx *ctx, double *theta);         /* gamma / sum_k gamma */
int tsamd_get_elogtheta(tsamd_ctx *ctx, double *elogtheta); /* psi(gamma) - psi(sum) */
int tsamd_set_counts(tsamd_ctx *ctx, const uint32_t *c);    /* _c_indiv, src/snpsamplinge.cc:18 */
int tsamd_get_counts(tsamd_ctx *ctx, uint32_t *c);

/* lambda[loc] is [k][2]; persists across visits to a location.  set also
 * refreshes Elogbeta (estimate_beta, src/snpsamplinge.cc:279-296). */
int tsamd_set_lambda(tsamd_ctx *ctx, uint32_t loc, const double *lambda);
int tsamd_get_lambda(tsamd_ctx *ctx, uint32_t first_loc, uint32_t n_locs, double *lambda /* [n_locs][k][2] */);
int tsamd_get_ebeta(tsamd_ctx *ctx, uint32_t first_loc, uint32_t n_locs, double *ebeta /* [n_locs][k] */);
int tsamd_get_elogbeta(tsamd_ctx *ctx, uint32_t first_loc, uint32_t n_locs, double *elogbeta /* [n_locs][k][2] */);

/* replaces one SNPSamplingE::optimize_lambda(loc) call (src/snpsamplinge.cc:320-366)
 * with the -nthreads 1 worker semantics (PhiRunnerE::do_work, :649-686): first the
 * pending gamma/Elogtheta step of the previous call is applied iff that call had
 * hol_mode == 0, using the phi of its LAST pass; then up to max_inner passes for loc;
 * then this call becomes the pending one.  inner_iters (may be NULL) = passes run. */
int tsamd_snp_update(tsamd_ctx *ctx, uint32_t loc, int hol_mode, uint32_t *inner_iters);
/* the same n times with no host round trip; asynchronous: returns after enqueueing, tsamd_synchronize() waits.
 * THE call to build on: on one GPU a whole schedule is ONE kernel launch whose weights never leave the registers
 * (13 600 updates/s at N = 1M, K = 8 against 7 750 for one tsamd_snp_update per update).  Results equal n x
 * tsamd_snp_update to rounding -- bit for bit in a given launch mode, except that a context in the default mode with
 * 4M or more weights per GPU (n x k) runs SINGLE-entry calls through the launch-per-SNP kernels, which are faster for
 * that shape and add the workgroups' partial sums in another order (rel 1e-11; TSAMD_SINGLE_ROUTE=0 disables). */
int tsamd_run_schedule(tsamd_ctx *ctx, const uint32_t *locs, uint32_t n, int hol_mode);
int tsamd_synchronize(tsamd_ctx *ctx);
/* optional: capture and instantiate, now, the hipGraphs tsamd_run_schedule replays (sequences of
 * 16, 8, 4, 2 and 1 SNPs; a schedule of any length is their binary decomposition), so that the
 * first tsamd_run_schedule call does not pay for it.  Otherwise that call does it.  On a sharded
 * context call it after the exchange has been set up (tsamd_comm_init / tsamd_p2p_connect). */
int tsamd_prepare(tsamd_ctx *ctx);
/* total inner passes executed by tsamd_run_schedule / snp_update calls since creation */
int tsamd_total_passes(tsamd_ctx *ctx, uint64_t *passes);
/* completed SNP updates since creation by the number of inner passes they ran: hist[i] for
 * i passes, i < TSAMD_PASS_HIST_BINS - 1; the last bin collects everything above.  (The bytes a
 * SNP-minibatch update moves depend on its pass count.) */
#define TSAMD_PASS_HIST_BINS 128
int tsamd_pass_histogram(tsamd_ctx *ctx, uint64_t hist[TSAMD_PASS_HIST_BINS]);
/* drop the pending gamma step (a new process starts with none: `first`, :652) */
int tsamd_clear_pending(tsamd_ctx *ctx);

/* replaces the second half of snp_likelihood (src/snpsamplinge.hh:336-360) for the
 * held-out individuals of loc in this shard: sum of log(max(C(2,y) q^y (1-q)^(2-y), 1e-30)),
 * q = sum_k Ebeta[loc][k] * Etheta[n][k]; ascending individual order. */
int tsamd_heldout_loglik(tsamd_ctx *ctx, uint32_t loc, double *sum, uint32_t *count);
/* replaces the validation block of compute_likelihood (src/snpsamplinge.cc:476-498:
 * for every validation location, snp_likelihood = optimize_lambda(loc) in hol mode, then the
 * held-out sum) for n locations at once.  theta does not move inside that block (only the
 * pending step of the last training SNP is applied, by the first update), so with run_updates != 0
 * this is one tsamd_run_schedule(locs, n, hol_mode = 1) followed by ONE kernel over all
 * (location, individual) entries of the context's device-resident held-out table -- no host
 * round trip and no allocation per location.  run_updates == 0 only evaluates (e.g. the initial
 * likelihood, :478 first == true; or after tsamd_run_schedule_all on several contexts).
 * loc_sums / loc_counts ([n], may be NULL) get what tsamd_heldout_loglik returns for each
 * location, bit for bit (entries are added in ascending individual order by one thread);
 * sum / count = their totals added in the listed order. */
int tsamd_heldout_eval(tsamd_ctx *ctx, const uint32_t *locs, uint32_t n, int run_updates, double *loc_sums,
                       uint32_t *loc_counts, double *sum, uint32_t *count);

/* ---- multi-GPU: individuals sharded, lambda_t all-reduced per pass over RCCL ----
 * rank 0 calls tsamd_comm_unique_id and ships the bytes to every rank (any
 * side channel, e.g. torch.distributed broadcast); every rank then calls
 * tsamd_comm_init with its context (cfg.rank / cfg.world).  Replaces the
 * main-thread reduction over workers (src/snpsamplinge.cc:337-352). */
int tsamd_comm_unique_id(uint8_t id[TSAMD_COMM_ID_BYTES]);
int tsamd_comm_init(tsamd_ctx *ctx, const uint8_t id[TSAMD_COMM_ID_BYTES]);

/* Peer-to-peer alternative to the RCCL all-reduce: each rank writes its 2K partial sums
 * straight into every peer's exchange buffer over xGMI (IPC-mapped fine-grained memory)
 * and the next pass adds the ranks' rows up in rank order, so the per-pass exchange costs
 * a few microseconds instead of a collective launch.  Every rank calls tsamd_p2p_export,
 * the handles of all ranks (rank order) are gathered over any side channel, every rank
 * calls tsamd_p2p_connect; a barrier is needed before the first update and before destroy.
 * All ranks must then issue identical call sequences (as with any collective).  A peer that
 * stops responding surfaces as TSAMD_ECOMM from tsamd_synchronize after a bounded wait. */
#define TSAMD_P2P_HANDLE_BYTES 64
int tsamd_p2p_export(tsamd_ctx *ctx, uint8_t handle[TSAMD_P2P_HANDLE_BYTES]);
int tsamd_p2p_connect(tsamd_ctx *ctx, const uint8_t *handles /* [world][TSAMD_P2P_HANDLE_BYTES] */);

/* The same exchange between contexts that live in ONE process (one context per GPU, ranks
 * 0 .. world-1 in any order in ctxs[]): no handles, peer access is enabled between the
 * devices.  The host then drives all shards from one thread the way the reference's main
 * thread drives its workers (src/snpsamplinge.cc:320-366): enqueue the same call on every
 * context (tsamd_run_schedule is asynchronous), then tsamd_synchronize each.  Calls that
 * synchronise internally (tsamd_snp_update) would wait for peers that have not been
 * enqueued yet: use tsamd_run_schedule_all + tsamd_synchronize in this mode.
 * (Two contexts on the SAME device -- a test set-up, pointless otherwise -- need a hardware
 * queue each: HIP maps streams onto GPU_MAX_HW_QUEUES queues, 4 by default, round-robin.) */
int tsamd_p2p_connect_local(tsamd_ctx *const *ctxs, uint32_t count);
/* tsamd_run_schedule on every context of such a group, interleaved in bounded batches: the
 * kernels of one shard wait for the other shards' kernels, so a thread must never queue an
 * unbounded amount of work on one context before the others have theirs (the device queue
 * would fill up and the thread block with the peers' work still unsubmitted).  Asynchronous
 * like tsamd_run_schedule; follow with tsamd_synchronize on each context. */
int tsamd_run_schedule_all(tsamd_ctx *const *ctxs, uint32_t count, const uint32_t *locs, uint32_t n, int hol_mode);

/* ---- measurement / synthetic workloads ---------------------------------------- */
/* Pritchard-Stephens-Donnelly genotypes straight into HBM for columns
 * [first_loc, first_loc + n_locs): y ~ Binomial(2, sum_k theta[n][k] * beta[j][k]),
 * theta [shard_count][k], beta [n_locs][k], counter-based RNG keyed by
 * (seed, global individual, location) so any sharding gives the same matrix.
 * missing_rate in [0,1) marks entries missing. */
int tsamd_synth_genotypes(tsamd_ctx *ctx, const double *theta, const double *beta,
                          uint32_t first_loc, uint32_t n_locs, uint64_t seed, double missing_rate);
/* HIP-event timing of the kernels issued by tsamd_run_schedule (eager launches while
 * enabled): per SNP one event pair brackets the first-pass kernel and one brackets the run
 * of max_inner-1 plain-pass launches.  Enable, run, then read launch counts and summed
 * bracket durations (a plain-pass bracket includes its inter-kernel gaps).  pass_launches
 * counts the plain passes that really swept (the near-empty launches after an early
 * convergence stay inside the bracket's time but are not counted). */
int tsamd_profile_enable(tsamd_ctx *ctx, int on);
int tsamd_profile_read(tsamd_ctx *ctx, uint64_t *pass_launches, double *pass_ms_total,
                       uint64_t *first_launches, double *first_ms_total);
/* On-box ceilings for the two pass kernels, measured on the context's own arrays and launch
 * geometry with the run's data resident: read_us = one bare streaming read of the weights (K
 * rows of 16-byte loads, alternating sweep direction: the plain pass' traffic minus the 2-bit
 * column), rmw_us = one bare read-modify-write of weights and gamma (the first pass' traffic;
 * the values are written back unchanged).  Averages over reps launches (HIP events).  The
 * wide-K fallback (k > TSAMD_SPECIALIZED_K) is probed with the same two kernels. */
int tsamd_probe_stream(tsamd_ctx *ctx, uint32_t reps, double *read_us, double *rmw_us);
/* how the context runs a SNP: kernels per SNP of the state-machine sequence (max_inner with one launch per
 * pass; 2 when all plain passes of a SNP run as one resident launch; 0 when a whole schedule runs as ONE launch with
 * the weights kept in registers), workgroups of the plain-pass and first-pass kernels.  The resident kernels need
 * k <= 32 and a shard that fits the register file of the GPU's compute units: 256 workgroups x
 *   k <= 8: 4096,  k = 9..16: 256 floor(128/k),  k = 17..24: 256 floor(112/k) (k = 22: 1024),  k = 25..32: 768   individuals
 * (1 048 576 per GPU at k <= 8, 524 288 at k = 16, 327 680 at k = 20; per RANK of a sharded run one 256-thread round less at
 * k = 14 and k = 16: 524 288 / 458 752); the whole-schedule kernel also nodekappa == 0.5.
 * A LARGER shard (k <= 32, nodekappa == 0.5, one GPU or up to 4 ranks connected peer to peer) still runs a whole schedule
 * as ONE launch (kernels_per_snp == 0): ts_hybrid keeps the weights of the first individuals of every thread in registers
 * and LDS and re-reads the others every pass (tsamd_schedule_geometry: on_chip_per_thread < indivs_per_thread); it has no
 * one-launch-per-SNP form (TSAMD_LAUNCH_PER_SNP is refused).  TSAMD_HYBRID=0 restores one launch per pass for such shards.
 * TSAMD_RESIDENT=0 / TSAMD_PERSISTENT=0 in the environment disable them. */
int tsamd_launch_info(tsamd_ctx *ctx, uint32_t *kernels_per_snp, uint32_t *plain_grid, uint32_t *first_grid);
/* Launch geometry of the context's resident kernel of `mode` (TSAMD_LAUNCH_PER_SNP: ts_resident, TSAMD_LAUNCH_PER_SCHEDULE:
 * ts_schedule; TSAMD_EUNSUPPORTED when the context does not qualify for it): workgroups (one per compute unit), individuals
 * per thread (each thread of a 256-thread workgroup holds that many individuals' weights in registers), and the levels of
 * the in-launch exchange -- 0: one workgroup, nothing is exchanged; 1: up to 32 (ts_resident: 16, k <= 8) workgroups on one
 * GPU, every workgroup reads every row; 2: groups of up to 32 workgroups and their leaders.  Small shards are launched on
 * fewer workgroups with more individuals per thread when that saves an exchange level.  on_chip_per_thread: how many of a
 * thread's individuals keep their weights on the chip for a whole launch (all of them in ts_schedule / ts_resident; a shard
 * above that register capacity runs TSAMD_LAUNCH_PER_SCHEDULE as ts_hybrid -- registers + LDS hold the weights of the first
 * ones, the others are re-read every pass -- and reports fewer than indivs_per_thread here).  Replaces the reference's
 * split_all_indivs bookkeeping (src/snpsamplinge.cc:298-318: nthreads chunks of floor(n / nthreads) individuals) as the
 * place where "who owns which individuals" is decided; tests assert the intended geometry through it. */
int tsamd_schedule_geometry(tsamd_ctx *ctx, int mode, uint32_t *workgroups, uint32_t *indivs_per_thread, uint32_t *exchange_levels,
                            uint32_t *on_chip_per_thread);
/* The batched validation block.  A validation-mode call (tsamd_run_schedule / tsamd_heldout_eval with hol_mode = 1) never
 * applies a gamma step between its entries (PhiRunnerE::do_work skips it under _prev_hol_mode, src/snpsamplinge.cc:660-668),
 * so theta is frozen and pairwise distinct locations are independent.  A context that runs TSAMD_LAUNCH_PER_SCHEDULE on one
 * GPU therefore runs such a call `batch` locations at a time (ts_holblock: one sweep of the register-resident weights per
 * sub-batch, ONE in-launch exchange per pass for the whole batch, the K x 2 epilogues side by side, per-location
 * convergence) -- after the call's first entry, which goes the ordinary way because it applies the pending gamma step of
 * the last training update.  Results equal the entry-by-entry path bit for bit (every per-location sum keeps its order).
 * Repeated locations cut the call into blocks of distinct ones.  A sharded context (tsamd_p2p_connect) that runs
 * TSAMD_LAUNCH_PER_SCHEDULE with ts_schedule does the same on every rank alike (the batch's rows are exchanged across the
 * ranks).  A context whose shard exceeds the register capacity (it runs ts_hybrid, one GPU or up to 4 ranks) batches the same
 * way with ts_hybhol: a sub-batch of locations shares one sweep of the weights -- registers + LDS + the streamed rest, read once
 * for the sub-batch -- and the batch one exchange per pass; bit for bit the entry-by-entry results as well.  batch = 0: the
 * context runs such calls entry by entry (other launch modes, TSAMD_HOLBLOCK=0 -- read per call; the same on every rank of a
 * sharded run).  launches / locations: ts_holblock / ts_hybhol launches so far and the entries they covered.  Replaces the loop
 * of compute_likelihood, src/snpsamplinge.cc:476-498. */
int tsamd_holblock_info(tsamd_ctx *ctx, uint32_t *batch, uint64_t *launches, uint64_t *locations);
/* Selects how the context launches from now on: one kernel per pass, one resident kernel for the plain passes of
 * a SNP, or one kernel per schedule.  tsamd_create picks the highest mode the context qualifies for; this call
 * can lower it and raise it again (TSAMD_EUNSUPPORTED above what the context qualifies for).  A sharded context
 * (tsamd_p2p_connect / _connect_local) qualifies for TSAMD_LAUNCH_PER_SCHEDULE when nodekappa == 0.5, world <= 8 and
 * every rank's shard fits as above and fills at least 8 workgroups (the in-launch exchange then spans the ranks; ranks
 * that share a device: TSAMD_DEVICE_SHARE=<ranks> in the environment), never for TSAMD_LAUNCH_PER_SNP.  All modes give the
 * same results to rounding (the order in which the workgroups' partial rows are added differs), and each mode is
 * bitwise reproducible and independent of how a schedule is cut into calls.  Synchronises the stream.
 *
 * The resident kernels exchange partial sums between their workgroups inside the launch and therefore need all of
 * them on the device at once.  Every such launch checks that first, with an empty exchange (bounded by
 * TSAMD_PROBE_MS, default 100 ms), before it modifies anything.  If something else holds compute units (another
 * context or process on the same GPU) the launch gives up with the state intact, every later kernel of the context
 * becomes a no-op, and the next synchronising call (tsamd_synchronize, a getter, ...) lowers the context to
 * TSAMD_LAUNCH_PER_PASS, replays the affected schedules from the unchanged state and returns success with a warning
 * in tsamd_last_error -- the results are those of an undisturbed run (to rounding, as between modes).  tsamd_recoveries
 * counts these events.  (The reference's counterpart: a run is never lost to its environment -- SIGTERM saves the
 * model, src/snpsamplinge.cc:454-457.)  A sharded context (one process per rank, tsamd_p2p_connect) does the same:
 * the entry exchange spans the ranks, so it fails on every rank with every rank's state intact, and every rank -- driven by
 * the same calls -- replays the same schedules one launch per pass; only the contexts of a tsamd_p2p_connect_local group
 * (one thread settles them one after the other) still report TSAMD_ECOMM.
 * The sharded recovery is BEST EFFORT.  The ranks agree on the entry's verdict through a second, empty exchange bounded by
 * a third of the wait of a regular one; a rank whose last workgroup gives up right at that deadline while its peers' pass
 * can still split the verdict (one rank replays, the others have begun to modify state and fail three exchanges later).
 * That case is never silent: every rank then ends with TSAMD_ECOMM (the replaying rank times out waiting for peers that
 * do not replay; a rank that went on finds the abort word and reports its state as void), and no rank continues on
 * diverged state.  Callers of a sharded run must therefore still handle TSAMD_ECOMM from any synchronising call. */
#define TSAMD_LAUNCH_PER_PASS 0
#define TSAMD_LAUNCH_PER_SNP 1      /* first pass + ts_resident */
#define TSAMD_LAUNCH_PER_SCHEDULE 2 /* ts_schedule */
int tsamd_set_launch_mode(tsamd_ctx *ctx, int mode);
int tsamd_recoveries(tsamd_ctx *ctx, uint32_t *count);
/* test / diagnostic aid: occupies `workgroups` compute units of the context's device for `milliseconds` with a
 * kernel on a second stream (returns once it runs) -- what another tenant of the GPU looks like to the resident kernels. */
int tsamd_debug_occupy(tsamd_ctx *ctx, uint32_t workgroups, uint32_t milliseconds);
/* device memory in bytes currently free / total on the context's device */
int tsamd_mem_info(tsamd_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);

#ifdef __cplusplus
}
#endif
#endif
