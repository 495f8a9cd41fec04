#!/usr/bin/env python3
"""SNP-minibatch updates/sec of the MI355X SVI engine (BASELINE.json metric).

One "step" = one SNP-minibatch update = tsamd_snp_update semantics
(optimize_lambda(loc), src/snpsamplinge.cc:320-366, plus the gamma/Elogtheta
step of that SNP, :695-740), on synthetic PSD genotypes resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
one rank per GPU; individuals are sharded across ranks and the per-pass
lambda statistics are exchanged inside libtsamd -- peer-to-peer stores over xGMI
when the start-up self-test passes, RCCL all-reduce otherwise (strong scaling:
N individuals fixed).  torch.distributed (gloo) only carries the bootstrap
handles and the timing barriers.

Everything the timed region needs is built before it starts whatever --warmup is
(tsamd_prepare: the replayed hipGraphs), and a schedule of any length replays without
padding, so `--steps 20 --warmup 5` measures the same per-update time as a long run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import rel_err, usable_cores  # noqa: E402  (tests/helpers.py: numpy only)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak (half the 157.3 TFLOP/s fp32 vector rate of MI355X_MICROARCH.md)
SELFTEST_TOL = 1e-9    # exchange self-test: lambda / gamma vs the CPU oracle after 6 updates (rel)


# One-GPU steady-state rates of the two BASELINE shapes that are sharded over a node (updates/s on ONE MI355X with everything resident,
# 2 000-update launches: profiles/r06_other_configs.txt) and what sharding is expected to buy: strong scaling of this path is poor by
# construction -- nine to ten DEPENDENT all-reduces per update, each a few microseconds of latency whatever the shard size.
ONE_GPU_STEADY = {(1_000_000, 8): 13850.0, (1_000_000, 20): 3550.0}
SCALING_NOTE = {
    8: ("strong scaling (N fixed); ten dependent exchanges per update; predicted 1.2 / 1.3 / 1.4 x the one-GPU rate at 2 / 4 / 8 GPUs for K = 8 "
        "(DESIGN.md section 5, predicted, never measured on a node): capacity, not speed, is what sharding buys this path"),
    20: ("strong scaling (N fixed); ten dependent exchanges per update; K = 20 at N = 1M exceeds one GPU's register file (half its weights are "
         "streamed there), so sharding also removes that stream: predicted 2.1 / 3.6 / 4.5 x the one-GPU rate at 2 / 4 / 8 GPUs "
         "(DESIGN.md section 5, predicted, never measured on a node)"),
    None: "strong scaling (N fixed); nine to ten dependent exchanges per update, each a few microseconds of latency whatever the shard size",
}


def resident_geometry(k):
    """(individuals per item, items per thread, items whose gamma stays in LDS) of the resident kernels --
    mirrors resident_vec / resident_items / sched_lds_items in csrc/tsamd_resident_kernels.h"""
    vec = 1
    items = 16 if k <= 8 else 128 // k if k <= 16 else 4 if k == 22 else 112 // k if k <= 24 else 3
    small = 1536 + 26 * 2 * k * 8
    lds = min(items, (160 * 1024 - small) // ((k * 8 + 4) * vec * 256))
    return vec, items, lds


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--individuals", "--n", dest="n", type=int, default=1_000_000, help="individuals (global)")
    ap.add_argument("--snps", "--l", dest="l", type=int, default=1_000_000,
                    help="SNP locations (capped to what fits in HBM)")
    ap.add_argument("--pops", "--k", dest="k", type=int, default=8)
    ap.add_argument("--seed", type=int, default=20240607)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all usable host cores (affinity and cgroup quota)")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event pass-kernel timing leg")
    ap.add_argument("--validation-locs", type=int, default=-1,
                    help="validation locations of the validation-block leg (-1: the reference's sample size, floor(0.005 L), at most 5000; 0: skip)")
    ap.add_argument("--ramp-seconds", type=float, default=0.3,
                    help="untimed priming schedule queued right ahead of the warm-up (set-up; 0 = none)")
    return ap.parse_args()


def fail_together(ok, dist, what):
    """Every rank leaves with a non-zero status if any rank failed (no rank is left waiting in a
    barrier for a peer that raised)."""
    if dist is not None:
        from terastructure_amd import dist as tdist

        ok = tdist.all_ok(ok, dist)
    if not ok:
        print(f"[bench] FAILED: {what}", file=sys.stderr, flush=True)
        if dist is not None:
            try:
                dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
        sys.exit(3)


def gather_bytes(local, dist, rank):
    """rank 0 gets every rank's byte array (list in rank order); others get None"""
    out = [None] * dist.get_world_size() if rank == 0 else None
    dist.gather_object(local, out, dst=0)
    return out


def choose_exchange(ts, dist, rank, world, local_rank, n, k, theta_shard, gamma_full, shard, cores):
    """Per-pass exchange of the 2K lambda statistics: direct peer-to-peer stores over xGMI or
    the RCCL all-reduce.  Both are run on this node on the benchmark's own shards (a few SNP
    columns, a short schedule) and checked (a) against the CPU oracle run by rank 0 on the same
    columns -- lambda, gamma and c_n after 6 updates -- and (b) against each other after 260.
    Peer-to-peer is used when it is valid and not slower; an exchange that fails its check is
    never timed.  TSAMD_EXCHANGE=rccl|p2p restricts the candidates (the check still runs)."""
    import torch

    import oracle_py as op
    from terastructure_amd import dist as tdist

    # candidates: the RCCL all-reduce, peer-to-peer stores with one launch per pass ("p2p"), and ts_schedule with its
    # in-launch exchange across the ranks ("p2p_schedule": one launch per rank and schedule; K <= 32 and shards that fit
    # the register file and fill 8 ... 256 workgroups)
    # ... and "p2p_schedule3": the same launch with THREE levels -- only the eight group leaders of a rank poll the world x 8
    # rows that arrive over xGMI and hand the total to their members (TSAMD_SCHEDULE_GATHER=leaders): one local hop more,
    # 32 x less polling of the fine-grained buffer.  Which of the two is faster on a node only the node can say.
    forced = os.environ.get("TSAMD_EXCHANGE", "auto").lower()
    all_modes = ["rccl", "p2p", "p2p_schedule", "p2p_schedule3"]
    modes = [forced] if forced in all_modes else all_modes
    l = 32
    sb, sc = shard
    beta = np.random.default_rng(7).uniform(0.05, 0.95, size=(l, k))
    locs = np.random.default_rng(8).integers(0, l, size=260).astype(np.uint32)
    locs[:6] = [3, 1, 3, 7, 0, 5]
    report = {"updates_per_s": {}, "rel_err_vs_oracle": {}, "valid": {}}
    final, want = {}, None
    def step(fn):
        """Run fn on this rank; (it succeeded on EVERY rank, its result here, its exception here).  Everything that can fail
        on one rank alone goes through this, so that all ranks always issue the same sequence of collectives: a rank that
        skipped a gather because ITS kernels timed out would leave the others waiting in it until gloo's timeout."""
        out, err = None, None
        try:
            out = fn()
        except Exception as exc:  # noqa: BLE001
            err = exc
        return tdist.all_ok(err is None, dist), out, err

    def note(mode, what, err):
        errs = [None] * world if rank == 0 else None
        dist.gather_object(None if err is None else f"{type(err).__name__}: {err}", errs, dst=0)
        if rank == 0:
            report.setdefault("errors", {})[mode] = {"what": what, "by_rank": {str(r): m for r, m in enumerate(errs) if m}}
            print(f"[bench] exchange self-test, {mode}: {what} failed on rank(s) "
                  f"{ {r: m for r, m in enumerate(errs) if m} }", file=sys.stderr, flush=True)

    for mode in modes:
        final[mode] = None
        report["valid"][mode] = False
        os.environ["TSAMD_SCHEDULE_GATHER"] = "leaders" if mode == "p2p_schedule3" else "all"   # (read when the exchange is connected)
        holder = {}

        def create():
            holder["e"] = ts.Engine(n, l, k, device=local_rank, rank=rank, world=world)
            holder["e"].synth_genotypes(theta_shard, beta, seed=11)
            holder["e"].set_gamma(gamma_full[sb:sb + sc])

        ok, _, err = step(create)
        e = holder.get("e")
        try:
            if not ok:
                note(mode, "set-up", err)
                continue
            try:
                (tdist.bootstrap_comm if mode == "rccl" else tdist.bootstrap_p2p)(e, dist)
            except Exception as exc:  # noqa: BLE001 -- raised on every rank together
                # (verbatim in the line: e.g. what RCCL said when it refused the communicator -- `rccl_error` for that candidate)
                report.setdefault("errors", {})[mode] = {"what": "bootstrap", "message": str(exc)}
                if mode == "rccl":
                    report["rccl_error"] = str(exc)
                if rank == 0:
                    print(f"[bench] exchange self-test, {mode}: {exc}", file=sys.stderr, flush=True)
                continue
            whole = e.launch_info()["kernels_per_snp"] == 0   # (the same on every rank: it depends on the configuration only)
            if mode == "p2p" and whole:
                e.set_launch_mode(ts.LAUNCH_PER_PASS)
            if mode.startswith("p2p_schedule") and not whole:
                report["valid"].pop(mode)
                continue                                       # the shards do not qualify: not a candidate
            if mode.startswith("p2p_schedule"):
                # which whole-schedule kernel the ranks run: ts_schedule (the shard's weights fit the register file) or, above
                # that capacity, ts_hybrid (registers + LDS + streamed weights; up to 4 ranks) -- the same on every rank
                try:
                    geo = e.schedule_geometry()
                    report.setdefault("schedule_kernel", {})[mode] = (
                        f"ts_hybrid<{k}> ({geo['on_chip_per_thread']} of {geo['indivs_per_thread']} individuals per thread on chip)"
                        if geo["indivs_per_thread"] > resident_geometry(k)[1] else f"ts_schedule<{k}> ({geo['indivs_per_thread']} individuals per thread)")
                except Exception:  # noqa: BLE001 -- a label only, the same on every rank
                    pass
            if want is None:  # the oracle's answer for the first 6 updates, once, on rank 0
                ok, cols, err = step(lambda: np.stack([e.download_bed(j) for j in range(l)]))          # [l][shard bytes]
                if not ok:
                    note(mode, "reading the columns back", err)
                    continue
                parts = gather_bytes(cols, dist, rank)

                def oracle():
                    if rank != 0:
                        return ()
                    orc = op.Oracle(n, l, k, nthreads=cores)
                    orc.load_bed_payload(np.concatenate(parts, axis=1)[:, :(n + 3) // 4])
                    orc.set_gamma(gamma_full)
                    for loc in locs[:6]:
                        orc.snp_update(int(loc))
                    out = (orc.lambda_(), orc.gamma(), orc.c_indiv())
                    orc.close()
                    return out

                ok, want_, err = step(oracle)
                if not ok:
                    note(mode, "the oracle", err)
                    continue
                want = want_
            res, dt, e6 = None, 0.0, float("inf")
            ok, st6, err = step(lambda: (e.run_schedule(locs[:6]), e.synchronize(), e.get_lambda(), e.get_gamma(),
                                         e.get_counts().astype(np.float64)[:, None])[2:])
            if ok:
                gam6 = tdist.gather_rows(st6[1], n, dist, ts.shard_range)
                cnt6 = tdist.gather_rows(st6[2], n, dist, ts.shard_range)[:, 0]
                if rank == 0:
                    e6 = max(rel_err(st6[0], want[0]), rel_err(gam6, want[1]))
                    if not np.array_equal(cnt6, want[2]):
                        e6 = float("inf")
                ok, _, err = step(lambda: (e.run_schedule(locs[6:60]), e.synchronize()))
            if ok:
                dist.barrier()

                def timed():
                    t0 = time.perf_counter()
                    e.run_schedule(locs[60:])
                    e.synchronize()
                    return time.perf_counter() - t0, (e.get_lambda(), e.get_gamma())

                ok, out, err = step(timed)
                if ok:
                    dt, res = out
            t6 = torch.tensor([e6 if rank == 0 else 0.0], dtype=torch.float64)
            dist.broadcast(t6, src=0)
            e6 = float(t6.item())
            report["rel_err_vs_oracle"][mode] = e6 if np.isfinite(e6) else None
            if ok:
                tt = torch.tensor([dt], dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                report["updates_per_s"][mode] = round((len(locs) - 60) / float(tt.item()), 1)
                # (to read the first real multi-GPU run against the single-GPU figures: DESIGN.md section 5 predicts the
                # per-pass cost of each candidate; one GPU: 6.1 us per pass inside ts_schedule, 3.1 of them in the exchange)
                report.setdefault("us_per_update", {})[mode] = round(float(tt.item()) / (len(locs) - 60) * 1e6, 2)
                report.setdefault("us_per_pass", {})[mode] = round(float(tt.item()) / (len(locs) - 60) * 1e6 / 10.0, 2)
                if e6 < SELFTEST_TOL:
                    final[mode] = res
                    report["valid"][mode] = True
            else:
                note(mode, "the run", err)
        finally:
            dist.barrier()
            if e is not None:
                e.close()
            dist.barrier()
    # candidates that matched the oracle after 6 updates must also agree with each other after 260
    ok = [m for m in modes if final.get(m) is not None]
    for i, a in enumerate(ok):
        for b in ok[i + 1:]:
            d = max(rel_err(x, y_) for x, y_ in zip(final[a], final[b]))
            td = torch.tensor([d], dtype=torch.float64)
            dist.all_reduce(td, op=dist.ReduceOp.MAX)
            report[f"{b}_vs_{a}_rel_err_260_updates"] = float(td.item())
            if float(td.item()) > 1e-8:  # they disagree although both matched the oracle early on: trust neither
                report["valid"][a] = report["valid"][b] = False
    rates = report["updates_per_s"]
    chosen = None
    for m in all_modes:   # (later ones win ties: fewer launches)
        if report["valid"].get(m) and (chosen is None or rates[m] >= rates[chosen]):
            chosen = m
    report["chosen"] = chosen
    if rank == 0:
        print(f"[bench] exchange self-test: {json.dumps(report)}", file=sys.stderr, flush=True)
    return chosen, report


def pmc_record_for(n, k, world, want, stale):
    """the committed counter record for this (N, K, GPUs, mode) -- only when it was collected from the kernels as they are
    now (hash of the device sources stored with the records, tools/pmc_record.py); stale[0] says why not otherwise"""
    pmc = os.path.join(ROOT, "profiles", "pass_kernel_pmc.json")
    try:
        doc = json.load(open(pmc))
        from terastructure_amd.build import kernel_sources_sha
        now, then = kernel_sources_sha(), doc.get("kernel_sources_sha")
        for rec in doc.get("records", []):
            if rec.get("n") == n and rec.get("k") == k and rec.get("n_gpus") == world and rec.get("mode", "pass") == want:
                if then != now:
                    stale[0] = (f"profiles/pass_kernel_pmc.json was collected from other kernel sources (sha {then}, the tree "
                                f"has {now}): its traffic / flops / latency figures are NOT used; re-profile "
                                "(tools/r06/profile.sh) to refresh them")
                    return {}
                return rec
    except Exception:  # noqa: BLE001
        pass
    return {}


def pass_kernels_roofline(pr, sub_mode, rec, k, sc, read_us, rmw_us):
    """roofline object of the launch-per-pass / launch-per-SNP kernels from a profile_read dict (HIP events around the first
    pass and around the plain passes of every SNP): HBM-bound, algorithmic bytes over the launch time.
    A plain pass is 8*N_shard*K (weights) + N_shard/4 (2-bit column) algorithmic bytes; the first pass (gamma step fused):
    R w, R gamma, W gamma, W w = 32*N*K; c_n R+W = 8N; two columns = N/2.  The resident kernel (mode "snp") runs ALL plain
    passes of a SNP in one launch that reads the weights once and keeps them in registers."""
    if not (pr["pass_launches"] and pr["first_launches"]):
        return None
    pass_bytes = 8.0 * sc * k + sc / 4.0
    first_bytes = 32.0 * sc * k + 8.0 * sc + sc / 2.0
    resident = sub_mode == "snp"
    first_s = pr["first_ms"] / pr["first_launches"] * 1e-3
    first_achieved = first_bytes / first_s / 1e9
    passes_per_launch = pr["pass_launches"] / pr["first_launches"] if resident else 1.0
    launches = pr["first_launches"] if resident else pr["pass_launches"]
    avg_s = pr["pass_ms"] / launches * 1e-3
    alg_bytes = passes_per_launch * pass_bytes
    equiv = None
    if resident:
        # the resident kernel reads the weights from memory ONCE per SNP and runs the later passes from registers:
        # its memory roofline is what it must move (weights once, one column), not passes x the plain-pass bytes
        equiv = {"bytes_per_launch": alg_bytes, "GBps": round(alg_bytes / avg_s / 1e9, 1),
                 "note": "passes x (8NK + N/4), the reference's dataflow, over this kernel's time: not a fraction of any peak"}
        alg_bytes = pass_bytes
        kernel = (f"ts_resident<{k}> (all {passes_per_launch:.3g} plain passes of a SNP in one launch: weights read once, "
                  "kept in registers; partial rows exchanged inside the launch)")
        note = ("achieved = the bytes the kernel must move per launch (the N x K weights once + one 2-bit column) over its "
                "launch time; `traffic` is the counter figure.  It is bound by neither memory nor arithmetic but by the "
                "in-launch exchange (about 3 us per pass with the ALU idle) plus the fp64 sweeps (2.6 us per pass at N = 1M, "
                "K = 8): per_pass_us x passes = avg_launch_us.  probe_read_us is a bare streaming read of the weights on this "
                "box (tsamd_probe_stream).")
    else:
        kernel = f"ts_pass<{k},false> (plain pass, max_inner - 1 launches per update)"
        note = ("fabric-side bandwidth incl. Infinity Cache, not DRAM bandwidth: the pass re-reads the same weights (8NK "
                "bytes: 64 MB at N=1M, K=8) every launch and they stay in the 256 MiB Infinity Cache; FETCH_SIZE counts "
                "those hits.  probe_read_us is a bare streaming read of the same array with the same geometry on this "
                "box (tsamd_probe_stream): the second denominator.")
    achieved = alg_bytes / avg_s / 1e9
    return {
        "bound": "hbm", "kernel": kernel,
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": rec.get("hbm_bytes_per_launch"),
        "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bandwidth_equiv": equiv,
        "avg_launch_us": round(avg_s * 1e6, 3), "launches_timed": launches,
        "passes_per_launch": round(passes_per_launch, 3),
        "per_pass_us": round(avg_s * 1e6 / passes_per_launch, 3),
        "ceiling_note": note,
        "probe_read_us": None if read_us is None else round(read_us, 3),
        "frac_of_probe": None if read_us is None else round(read_us * 1e-6 / avg_s, 4),
        "first_pass": {
            "kernel": f"ts_pass<{k},true> (first pass of a SNP + the previous SNP's gamma step, 1 launch per update)",
            "bound": "hbm", "achieved": round(first_achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(first_achieved / HBM_PEAK_GBS, 4), "traffic": rec.get("first_pass_hbm_bytes_per_launch"),
            "algorithmic_bytes_per_launch": first_bytes, "avg_launch_us": round(first_s * 1e6, 3),
            "launches_timed": pr["first_launches"],
            "probe_rmw_us": None if rmw_us is None else round(rmw_us, 3),
            "frac_of_probe": None if rmw_us is None else round(rmw_us * 1e-6 / first_s, 4),
        },
    }


def schedule_roofline(prs, ran, nsteps, rec, stale, k, sc, geo, read_us):
    """roofline object of the whole-schedule kernel (ts_schedule, or ts_hybrid above the register capacity) from the HIP
    events around its launches (prs), the passes the device ran (ran) over nsteps updates, and -- when one is committed
    for this shape -- the counter record rec.  sc = individuals of this rank's shard; figures are per GPU."""
    if not prs["pass_launches"]:
        return None
    pass_bytes = 8.0 * sc * k + sc / 4.0
    first_bytes = 32.0 * sc * k + 8.0 * sc + sc / 2.0
    launch_s = prs["pass_ms"] / prs["pass_launches"] * 1e-3
    upd = nsteps / prs["pass_launches"]
    ppu = ran / nsteps                                   # passes per update
    # (1) what binds it: fp64 vector arithmetic at one wave per SIMD.  Flops per update: the count of the kernel's own
    # formulation (FMA = 2): a sweep is 8K + 12 per individual (two K-term normalisers, ONE reciprocal of their
    # product with its third-order step, 2K accumulations per parent), the gamma step 89K + 25 (normalisers 4K,
    # update 7K -- 10K in the full-size K <= 8 instantiation, which keeps the literal form --, exp(psi) 78K).
    literal_step = k <= 8 and sc > 15 * 65536
    hand = ppu * sc * (8.0 * k + 12.0) + sc * ((92.0 if literal_step else 89.0) * k + 25.0)
    # `achieved` / `frac` use the ALGORITHMIC count (the formulation's flops, computed here from this run's pass counts).
    # The counters of a profiled launch (when a record for this shape is committed) give what the kernel EXECUTED, reported
    # beside it: wave instructions x 64 lanes, which also prices the K x 2 epilogues -- 2K active lanes of a wave, run by all
    # four waves of all workgroups -- at full width (18 % above the algorithmic count at N = 1M, K = 8).
    flops = hand
    flops_src = ("algorithmic: the formulation's flops, FMA = 2 -- per update passes x N x (8K + 12) for the sweeps + N x (89K + 25) "
                 "for the gamma step (92K in the full-size K <= 8 instantiation), from THIS run's pass count")
    executed = rec.get("fp64_flops_per_update")
    executed_src = None
    if executed:
        executed_src = ("SQ_INSTS_VALU_*_F64 counters of a profiled launch committed under profiles/ (constants of "
                        "profiles/pass_kernel_pmc.json, guarded by a hash of the kernel sources -- not measured in this run): "
                        + ", ".join(rec.get("flops_source_files", [])))
    elif stale[0]:
        executed_src = stale[0]
    tflops = flops * upd / launch_s / 1e12
    # (2) memory: what the kernel itself must move per update -- gamma and c_n, read and written, of the items
    # whose gamma is not kept in LDS, one 2-bit column -- and what the counters saw
    vec, items, n_lds = resident_geometry(k)
    moved = (16.0 * sc * k + 8.0 * sc) * (items - n_lds) / items + sc / 4.0
    traffic = rec.get("hbm_bytes_per_update")
    # (3) the reference's dataflow: per update one first pass 32NK + 8N + N/2 and passes - 1 plain
    # passes 8NK + N/4 -- what this kernel would have to move if the weights did not stay in registers
    alg_bytes = (nsteps * first_bytes + max(0, ran - nsteps) * pass_bytes) / prs["pass_launches"]
    # (4) latency: the in-launch exchanges, during which the vector ALU idles (in-kernel timers of the
    # diagnostic build -DTSAMD_SCHED_TIME, recorded with the counters)
    xus = rec.get("exchange_us_per_update")
    roofline = {
        "bound": "fp64_valu",
        "kernel": (f"ts_schedule<{k}> (one launch = {upd:.0f} SNP updates: the gamma step and all {ppu:.3g} passes of every "
                   "SNP; weights in registers from the first SNP to the last; partial rows exchanged inside the launch)"),
        "achieved": round(tflops, 2), "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(tflops / FP64_VALU_PEAK_TFLOPS, 4),
        "traffic": None if traffic is None else traffic * upd,
        "flops_per_update": flops, "flops_source": flops_src,
        "executed": {"flops_per_update": executed, "achieved": None if not executed else round(executed * upd / launch_s / 1e12, 2),
                     "frac": None if not executed else round(executed * upd / launch_s / 1e12 / FP64_VALU_PEAK_TFLOPS, 4),
                     "unit": "TFLOP/s", "source": executed_src},
        "measured_in_this_run": "avg_launch_us (HIP events on the engine's stream around every launch of the timed kernel)",
        "avg_launch_us": round(launch_s * 1e6, 1), "launches_timed": prs["pass_launches"], "updates_per_launch": upd,
        "per_update_us": round(launch_s * 1e6 / upd, 3), "passes_per_update": round(ppu, 3),
        "bound_note": ("fp64 vector issue: the kernel runs one wave per SIMD (a thread owns the whole register file), where "
                       "tools/fma_probe reaches 62.5 of the 78.6 TFLOP/s; the rest of the distance is the exchange latency "
                       "(`latency`) and instructions that are not flops (register moves between the AGPR-resident weights "
                       "and the ALU, code decode): every vector instruction costs the lone wave 4.3-4.7 cycles, "
                       "tools/ubench/op_cost.hip.  HBM is far from binding (`hbm`)."),
        "hbm": {
            "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "achieved": None if traffic is None else round(traffic * upd / launch_s / 1e9, 1),
            "frac": None if traffic is None else round(traffic * upd / launch_s / 1e9 / HBM_PEAK_GBS, 4),
            "traffic_bytes_per_update": traffic,
            "moved_bytes_per_update": moved, "moved_GBps": round(moved * upd / launch_s / 1e9, 1),
            "moved_frac": round(moved * upd / launch_s / 1e9 / HBM_PEAK_GBS, 4),
            "note": ("achieved / frac: FETCH_SIZE x 2 + WRITE_SIZE of a profiled launch (profiles/pass_kernel_pmc.json) over "
                     "this run's launch time; moved_*: the bytes the kernel must move by construction (streamed gamma "
                     "read + write, c_n, one 2-bit column)"),
        },
        "algorithmic_bandwidth_equiv": {
            "bytes_per_update": alg_bytes / upd, "GBps": round(alg_bytes / launch_s / 1e9, 1),
            "note": ("SURVEY 8(d) bytes of the reference's dataflow (every pass re-reads the N x K weights) over this kernel's "
                     "time: a speed-up figure against a memory-bound implementation, not a fraction of any peak -- the "
                     "kernel does not move these bytes"),
        },
        "latency": {
            "exchanges_per_update": round(ppu, 3),
            "exchange_us_per_update": xus,
            "frac_of_update": None if xus is None else round(xus / (launch_s * 1e6 / upd), 4),
            "source": rec.get("exchange_source", "no in-kernel timer record for this N, K"),
        },
        "launch_per_snp": None, "first_pass": None,
        "probe_read_us": None if read_us is None else round(read_us, 3),
    }
    # A shard above ts_schedule's register capacity runs the same one-launch structure as ts_hybrid: part of the
    # weights in registers + LDS, the rest re-read from memory every pass, all of gamma streamed -- that kernel is
    # bound by memory, and priced so: the bytes it must move by construction over the launch time against the HBM peak.
    if geo and geo["indivs_per_thread"] > items:
        on_chip = min(sc, geo["workgroups"] * 256 * geo["on_chip_per_thread"])
        streamed = sc - on_chip
        moved_h = ppu * streamed * 8.0 * k + sc * (16.0 * k + 8.0) + streamed * 16.0 * k + (ppu + 1.0) * sc / 4.0
        kernel_h = (f"ts_hybrid<{k}> (one launch = {upd:.0f} SNP updates; of a thread's {geo['indivs_per_thread']} individuals "
                    f"{geo['on_chip_per_thread']} keep their weights in registers + LDS for the whole launch, the weights of the "
                    "others are re-read every pass; gamma and c_n of all stream through the gamma step)")
        hbm_h = {"bound": "hbm", "achieved": round(moved_h * upd / launch_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(moved_h * upd / launch_s / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_update": moved_h,
                 "algorithmic_bytes_note": ("passes x streamed individuals x 8K (weights re-read) + N (16K + 8) (gamma, c_n read and "
                                            "written) + streamed x 16K (their weights read and written by the gamma step) + "
                                            "(passes + 1) N / 4 (columns)"),
                 "streamed_individuals": int(streamed), "on_chip_individuals": int(on_chip)}
        if streamed > 0:   # memory binds: the streamed weights and the gamma step's streams
            fp64 = {key: roofline[key] for key in ("achieved", "peak", "unit", "frac", "flops_per_update", "flops_source", "executed")}
            fp64["bound"] = "fp64_valu"
            roofline.update(hbm_h)
            roofline.update({
                "kernel": kernel_h, "fp64_valu": fp64,
                "bound_note": ("memory: the streamed weights (Infinity Cache / HBM) and the gamma step's streams; the exchanges and "
                               "epilogues (`latency`) run with the memory system idle"),
            })
        else:              # everything on chip: like ts_schedule, with all of gamma streamed through the gamma step
            roofline["kernel"] = kernel_h
            roofline["hbm"] = hbm_h
        roofline["traffic"] = None if not rec.get("hbm_bytes_per_update") else rec["hbm_bytes_per_update"] * upd
    return roofline


def measure_legs(args, ts, eng, dist, rank, world, local_rank, n, l, k, sc, locs, cores, oracle_ok, gamma_init):
    """Everything the JSON line carries besides `value`, measured after the timed region: the roofline of the TIMED kernel
    (HIP events around its launches, in the mode that was timed -- no mode switch), then the secondary legs -- the kernels
    of the other launch modes, the device-copy ceiling, the validation block, the CPU baseline and the GPU's parity with it.
    Every leg runs in its own try: one that fails (a peer timing out in a rehearsal with all ranks on one GPU, a mode the
    context does not qualify for) is recorded in `legs` and cannot null the others.  N > 1: every rank runs every leg, and a
    leg counts as done only if it succeeded on EVERY rank (one all_ok collective per leg keeps the ranks aligned); after a
    failed leg that ran kernels of a sharded context -- whose state may be void then -- the remaining GPU legs are skipped."""
    from terastructure_amd import dist as tdist

    legs = {}
    usable = [True]   # the engine is still good for GPU legs

    def leg(name, fn, gpu=True):
        if gpu and not usable[0]:
            legs[name] = "skipped: an earlier leg left the context unusable"
            return None
        out, err = None, None
        try:
            out = fn()
        except Exception as exc:  # noqa: BLE001
            err = exc
        ok = err is None
        if dist is not None:
            ok = tdist.all_ok(ok, dist)
        if ok:
            legs[name] = "ok"
            return out
        legs[name] = f"failed: {type(err).__name__}: {err}" if err is not None else "failed on a peer rank"
        print(f"[bench] leg {name} {legs[name]}; the line is reported without it", file=sys.stderr, flush=True)
        if gpu:
            if dist is not None:
                usable[0] = False   # (a sharded context: the peers' states may have diverged -- no further kernels)
            else:
                try:
                    eng.synchronize()
                except Exception:  # noqa: BLE001
                    usable[0] = False
        return None

    info = eng.launch_info()
    kps = info["kernels_per_snp"]
    mode = "schedule" if kps == 0 else "snp" if (kps == 2 and eng.cfg.max_inner > 2) else "pass"
    stale = [None]  # why the committed counter records were not used (None: they were, or there are none for this shape)
    warm = args.warmup

    def profiled(nsteps):
        """(profile_read dict, passes the device ran) over nsteps updates"""
        eng.synchronize()
        q0 = eng.total_passes()
        eng.profile_enable(True)
        try:
            eng.run_schedule(locs[warm:warm + nsteps])
            eng.synchronize()
            pr_ = eng.profile_read()
        finally:
            eng.profile_enable(False)
        return pr_, eng.total_passes() - q0

    roofline = None
    read_us = rmw_us = None
    if not args.no_profile:
        probe = leg("stream_probe", lambda: eng.probe_stream(50), gpu=False)   # (rank-local kernels: no peer is involved)
        if probe is not None:
            read_us, rmw_us = probe

        # ---- the timed kernel, in the mode that was timed -----------------------------------
        def timed_kernel():
            if mode == "schedule":
                nsteps = min(args.steps, 2000)
                prs, ran = profiled(nsteps)
                try:
                    geo = eng.schedule_geometry()
                except Exception:  # noqa: BLE001
                    geo = None
                return schedule_roofline(prs, ran, nsteps, pmc_record_for(n, k, world, "schedule", stale), stale, k, sc, geo, read_us)
            pr, _ = profiled(min(args.steps, 300))
            return pass_kernels_roofline(pr, mode, pmc_record_for(n, k, world, mode, stale), k, sc, read_us, rmw_us)

        roofline = leg("roofline_timed_kernel", timed_kernel)

        # ---- secondary: the kernels of the launch-per-SNP / launch-per-pass sequence, one GPU only (a sharded context is
        # not switched between modes for a measurement: its other sequence waits on peers with its own bounded protocol)
        if roofline is not None and mode == "schedule" and world == 1:
            def other_modes():
                sub = ts.LAUNCH_PER_SNP
                try:
                    eng.set_launch_mode(sub)
                except ts.TsamdError:       # (a shard above the register capacity -- ts_hybrid -- has no launch-per-SNP mode)
                    sub = ts.LAUNCH_PER_PASS
                    eng.set_launch_mode(sub)
                try:
                    pr, _ = profiled(min(args.steps, 300))
                finally:
                    eng.set_launch_mode(ts.LAUNCH_PER_SCHEDULE)
                sub_mode = "snp" if (sub == ts.LAUNCH_PER_SNP and eng.cfg.max_inner > 2) else "pass"
                return pass_kernels_roofline(pr, sub_mode, pmc_record_for(n, k, world, sub_mode, stale), k, sc, read_us, rmw_us)

            per_snp = leg("roofline_other_launch_modes", other_modes)
            if per_snp is not None:
                roofline["launch_per_snp"] = per_snp
                roofline["first_pass"] = per_snp["first_pass"]
        if roofline is not None:
            roofline["counter_records"] = stale[0] or ("profiles/pass_kernel_pmc.json matches the kernel sources of this tree (or holds no "
                                                       "record for this shape)")

        # third denominator: what a plain device-to-device copy reaches on this box,
        # with the benchmark's data still resident (read + write bytes over the copy time)
        def device_copy():
            import torch

            dev = torch.device("cuda", local_rank)
            src = torch.empty(1 << 27, dtype=torch.float64, device=dev)  # 1 GiB
            dst = torch.empty_like(src)
            src.zero_()
            for _ in range(2):
                dst.copy_(src)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(10):
                dst.copy_(src)
            ev1.record()
            torch.cuda.synchronize(dev)
            return round(10 * 2 * src.numel() * 8 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9, 1)

        if roofline is not None:
            roofline["device_copy_GBps"] = leg("device_copy", device_copy, gpu=False)

    # ---- CPU baseline: the oracle ("port") on the host cores of rank 0, bounded sample; then the GPU(s) ----
    # ---- repeat exactly those updates from the same start and the two states are compared ----
    cpu, parity = None, None
    if args.cpu_seconds > 0:
        ls = 8  # sample columns; per-update cost does not depend on L
        box = {}

        # RULE of this function: a leg's body does only work that can fail on ONE rank alone (kernels, copies, the oracle);
        # torch.distributed collectives stand OUTSIDE the legs, after the leg's own all_ok -- every rank reaches them or none.
        cols = leg("sample_columns", lambda: np.stack([eng.download_bed(int(j)) for j in range(ls)]), gpu=False)
        got = None
        if cols is not None:
            # (the shards' byte ranges of a column are contiguous slices of the .bed column: tsamd_shard_range)
            if dist is not None:
                parts = gather_bytes(cols, dist, rank)
                cols = np.concatenate(parts, axis=1)[:, :(n + 3) // 4] if rank == 0 else None

            def cpu_baseline():
                if rank != 0:
                    return None
                if not oracle_ok:
                    raise RuntimeError("oracle/libts_oracle.so is not available")
                import oracle_py as op

                g0 = gamma_init()

                def run_oracle(threads, budget, max_updates):
                    orc = op.Oracle(n, ls, k, nthreads=threads, gamma_scale=float(l))
                    orc.load_bed_payload(cols)
                    orc.set_gamma(g0)
                    seq = [0]
                    orc.snp_update(0)  # untimed: the first call has no gamma step to apply
                    done, tc0 = 0, time.perf_counter()
                    while True:
                        seq.append((done + 1) % ls)
                        orc.snp_update(seq[-1])
                        done += 1
                        if time.perf_counter() - tc0 > budget or done >= max_updates:
                            break
                    return orc, seq, done, time.perf_counter() - tc0

                orc, seq, done, cdt = run_oracle(cores, args.cpu_seconds * 0.75, args.steps)
                box["want"] = (orc.lambda_(), orc.gamma(), orc.c_indiv())
                orc.close()
                orc1, _, done1, cdt1 = run_oracle(1, args.cpu_seconds * 0.25, 4)
                orc1.close()
                return (seq, {"value": round(done / cdt, 4), "unit": "SNP-minibatch updates/s", "cores": cores,
                              "kind": "port", "value_1_thread": round(done1 / cdt1, 4),
                              "sample": f"{done} updates (10 passes + gamma step each) at N={n}, K={k} on {ls} of the "
                                        f"benchmark's own columns, oracle/ts_oracle.c with {cores} OpenMP threads "
                                        f"in the reference's work partition on rank 0's host (it shows {os.cpu_count()} CPUs, "
                                        f"{cores} usable under its affinity mask / cgroup quota); value_1_thread: "
                                        f"{done1} updates with one thread"})

            got = leg("cpu_baseline", cpu_baseline, gpu=False)
            if legs["cpu_baseline"] == "ok" and dist is not None:   # (the other ranks need the sequence the oracle ran)
                res = [got]
                dist.broadcast_object_list(res, src=0)
                got = res[0]
        if got is not None:
            seq, cpu = got
            # the same updates on the GPU(s), from the same state (lambda of the sample columns back to eta, gamma and c_n
            # back to the start, no pending step), through the timed entry point -- in the mode that was timed and, on one
            # GPU, in every other launch mode whose kernels this line publishes timings of
            eta = np.ones((k, 2))
            sb = eng.shard_begin

            def gpu_repeat():
                for j in range(ls):
                    eng.set_lambda(j, eta)
                eng.set_gamma(gamma_init()[sb:sb + sc])
                eng.set_counts(np.zeros(sc, dtype=np.uint32))
                eng.clear_pending()
                eng.run_schedule(np.array(seq, dtype=np.uint32))
                eng.synchronize()
                return eng.get_lambda(0, ls), eng.get_gamma(), eng.get_counts()

            def compare(state):
                """rank 0: the (gathered) state against the oracle's"""
                lam, gam, cnt = state
                if dist is not None:
                    gam = tdist.gather_rows(gam, n, dist, ts.shard_range)
                    cnt = tdist.gather_rows(cnt.astype(np.float64)[:, None], n, dist, ts.shard_range)[:, 0]
                if rank != 0:
                    return {}
                want = box["want"]
                e_lam, e_gam = rel_err(lam, want[0]), rel_err(gam, want[1])
                cnt_eq = bool(np.array_equal(cnt, want[2]))
                return {"lambda_rel_err": e_lam, "gamma_rel_err": e_gam, "c_n_equal": cnt_eq,
                        "ok": bool(e_lam < 1e-9 and e_gam < 1e-9 and cnt_eq)}

            state = leg("parity_vs_cpu_baseline", gpu_repeat)
            if state is not None:
                parity = compare(state)
                timed_kps = eng.launch_info()["kernels_per_snp"]
                parity.update({"updates": len(seq), "tolerance": 1e-9, "kernels_per_snp": timed_kps})
                if world == 1:
                    def other_modes_parity():
                        others = {}
                        try:
                            for name, m_ in (("launch_per_snp", ts.LAUNCH_PER_SNP), ("launch_per_pass", ts.LAUNCH_PER_PASS)):
                                try:
                                    eng.set_launch_mode(m_)
                                except ts.TsamdError:
                                    continue                                  # the context does not qualify for it
                                if eng.launch_info()["kernels_per_snp"] != timed_kps:
                                    others[name] = compare(gpu_repeat())
                                    others[name]["kernels_per_snp"] = eng.launch_info()["kernels_per_snp"]
                        finally:
                            try:
                                eng.set_launch_mode(ts.LAUNCH_PER_SCHEDULE)
                            except ts.TsamdError:
                                pass
                        return others

                    others = leg("parity_other_launch_modes", other_modes_parity)
                    if others:
                        parity["other_launch_modes"] = others
                        parity["ok"] = bool(parity["ok"] and all(o["ok"] for o in others.values()))
    # ---- the validation block (compute_likelihood, src/snpsamplinge.cc:461-544) at this configuration: the reference's sample
    # (floor(0.005 L) locations x N/100 held-out individuals, src/snpsamplinge.cc:196-224) registered through tsamd_set_heldout,
    # reports through tsamd_heldout_eval -- batched (ts_holblock) and, for comparison, entry by entry (TSAMD_HOLBLOCK=0).
    # Not part of `value`: a report runs once per -rfreq training updates.  N > 1: every rank registers the same sample (global
    # individual ids; a rank keeps its own), the ranks' sums and counts are added.
    validation = None
    nval = min(5000, l // 200) if args.validation_locs < 0 else min(args.validation_locs, l - 8)
    if nval >= 2 and not args.no_profile:
        def validation_block():
            if world == 1:
                try:
                    eng.set_launch_mode(ts.LAUNCH_PER_SCHEDULE)
                except ts.TsamdError:
                    pass
            vrng = np.random.default_rng(args.seed + 77)
            vlocs = np.sort(8 + vrng.choice(l - 8, size=nval, replace=False)).astype(np.uint32)
            per_loc = n // 10 if n < 2000 else n // 100
            tv0 = time.perf_counter()
            for loc in vlocs:
                eng.set_heldout(int(loc), np.sort(vrng.choice(n, size=per_loc, replace=False)).astype(np.uint32))
            t_set = time.perf_counter() - tv0
            train = vrng.integers(0, l, size=64).astype(np.uint32)

            def report():
                eng.run_schedule(train)        # (a report follows training: its first entry applies the pending gamma step)
                eng.synchronize()
                tq = time.perf_counter()
                sq, cq, _, _ = eng.heldout_eval(vlocs, run_updates=True)
                return time.perf_counter() - tq, sq, int(cq)

            report()
            batched = sorted(report() for _ in range(3))[1]
            hinfo = eng.holblock_info()
            os.environ["TSAMD_HOLBLOCK"] = "0"
            try:
                single = report()
            finally:
                del os.environ["TSAMD_HOLBLOCK"]
            tq = time.perf_counter()
            eng.heldout_eval(vlocs, run_updates=False)
            t_eval = time.perf_counter() - tq
            try:   # (which kernel batches: ts_holblock beside ts_schedule, ts_hybhol beside ts_hybrid -- a shard above the register capacity)
                geo = eng.schedule_geometry()
                hyb = geo["indivs_per_thread"] > resident_geometry(k)[1]
            except Exception:  # noqa: BLE001 -- a label only
                hyb = False
            return dict(batched=batched, single=single, t_eval=t_eval, t_set=t_set, per_loc=per_loc, batch=hinfo["batch"], hybrid=hyb)

        v = leg("validation_block", validation_block)
        if v is not None:
            batched, single = v["batched"], v["single"]
            if dist is not None:   # this shard's sums / counts / wall times -> the run's
                sq, cq = tdist.sum_over_ranks([batched[1], float(batched[2])], dist)
                batched = (tdist.max_over_ranks(batched[0], dist), sq, int(cq))
                single = (tdist.max_over_ranks(single[0], dist), single[1], single[2])
            validation = {
                "locations": int(nval), "heldout_per_location": int(v["per_loc"]), "heldout_entries": int(batched[2]),
                "kernel": ((f"ts_hybhol<{k}>: {v['batch']} locations per exchange, sub-batches share one sweep of the streamed weights" if v["hybrid"]
                            else f"ts_holblock<{k}>: {v['batch']} locations per sweep group and exchange") if v["batch"] else
                           "entry by entry (the context does not run a batched validation kernel)"),
                "seconds_per_report": round(batched[0], 4), "us_per_location": round(batched[0] / nval * 1e6, 2),
                "entry_by_entry_seconds_per_report": round(single[0], 4),
                "entry_by_entry_us_per_location": round(single[0] / nval * 1e6, 2),
                "evaluation_only_seconds": round(v["t_eval"], 4),
                "mean_loglik": round(batched[1] / max(1, batched[2]), 6),
                "set_heldout_seconds": round(v["t_set"], 2),
                "note": ("one report = hol-mode updates of all validation locations (theta frozen: batched) + the held-out "
                         "log-likelihood of all entries; wall time of tsamd_heldout_eval (max over ranks), median of three"),
            }

    return {"roofline": roofline, "cpu_baseline": cpu, "parity": parity if rank == 0 else None, "validation_block": validation, "legs": legs}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "TSAMD_BENCH_DEVICE" in os.environ:  # development only: several ranks on one GPU
        local_rank = int(os.environ["TSAMD_BENCH_DEVICE"])
        os.environ.setdefault("TSAMD_DEVICE_SHARE", str(world))  # their resident kernels must fit that GPU together
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    # The oracle (the checker: cpu_baseline leg, exchange self-test) is loaded -- and built if the
    # library is missing -- BEFORE anything initialises the GPU: no child process may be started
    # from a process that has.
    cores = args.cpu_threads or usable_cores()
    oracle_ok = True
    if rank == 0 and (world > 1 or args.cpu_seconds > 0):
        try:
            import oracle_py as op

            op.lib()
        except Exception as exc:  # noqa: BLE001
            oracle_ok = False
            print(f"[bench] oracle library unavailable ({exc})", file=sys.stderr, flush=True)

    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (the longest legitimate gap between two collectives is rank 0 running the oracle for the self-test: seconds)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=6))
        fail_together(oracle_ok, dist, "the exchange self-test needs oracle/libts_oracle.so on rank 0")

    import torch

    import terastructure_amd as ts

    n, k = args.n, args.k
    sb, sc = ts.shard_range(n, rank, world)
    colstride = ((sc + 511) // 512 * 512) // 4

    # how many SNP columns fit next to the per-individual and per-location state
    probe = ts.Engine(min(n, 4096), 16, k, device=local_rank)
    free_b, total_b = probe.mem_info()
    probe.close()
    per_loc = colstride + 2 * (2 * k * 8)          # column + lambda + exp(Elogbeta)
    reserve = 4 * (sc * k * 8) + (6 << 30)          # w, gamma, synth scratch + headroom
    l = int(min(args.l, max(64, (free_b - reserve) // per_loc)))
    if dist is not None:
        t = torch.tensor([l], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        l = int(t.item())

    t_setup = time.time()
    # synthetic PSD data: theta ~ Dir(0.2), beta ~ U(0.05, 0.95), y ~ Bin(2, theta.beta)
    rng = np.random.default_rng(args.seed)
    theta = rng.dirichlet(np.full(k, 0.2), size=n)[sb:sb + sc]

    def gamma_init():
        return np.random.default_rng(args.seed + 2).gamma(100.0, 0.01, size=(n, k))

    gamma0 = gamma_init()
    exchange, exchange_report = "none", {}
    if world > 1:
        exchange, exchange_report = choose_exchange(ts, dist, rank, world, local_rank, n, k, theta, gamma0, (sb, sc),
                                                    cores)
        # an exchange that did not reproduce the oracle is never timed
        fail_together(exchange is not None, dist,
                      f"no exchange passed its self-test on this node: {json.dumps(exchange_report)}")
    # (every step that can fail on one rank alone ends in fail_together: no rank is left in a collective a dead peer skipped)
    eng, err = None, None
    try:
        eng = ts.Engine(n, l, k, device=local_rank, rank=rank, world=world)
    except Exception as exc:  # noqa: BLE001
        err = exc
    fail_together(err is None, dist, f"engine creation: {err}")
    if world > 1:
        from terastructure_amd import dist as tdist

        try:
            os.environ["TSAMD_SCHEDULE_GATHER"] = "leaders" if exchange == "p2p_schedule3" else "all"
            (tdist.bootstrap_comm if exchange == "rccl" else tdist.bootstrap_p2p)(eng, dist)
            if exchange == "p2p" and eng.launch_info()["kernels_per_snp"] == 0:
                eng.set_launch_mode(ts.LAUNCH_PER_PASS)   # (the in-launch exchange did not pass, or was slower)
        except Exception as exc:  # noqa: BLE001
            err = exc
        fail_together(err is None, dist, f"exchange bootstrap: {err}")

    try:
        chunk = 1 << 17
        brng = np.random.default_rng(args.seed + 1)
        for l0 in range(0, l, chunk):
            beta = brng.uniform(0.05, 0.95, size=(min(chunk, l - l0), k))
            eng.synth_genotypes(theta, beta, first_loc=l0, seed=args.seed)
        eng.set_gamma(gamma0[sb:sb + sc])
        del theta, gamma0
        eng.prepare()  # graphs captured + instantiated here, not inside the timed region
    except Exception as exc:  # noqa: BLE001
        err = exc
    fail_together(err is None, dist, f"set-up of the benchmark's data: {err}")
    try:
        # bring the device to its working clocks before the (possibly very short) warm-up: ~20 ms of the
        # bare streaming probes, which read the weights and write them back unchanged
        eng.probe_stream(400)
    except Exception:  # noqa: BLE001 -- not essential
        pass
    # Untimed priming schedule (reported as setup.clock_ramp_updates).  The chip's power management slows a launch that
    # follows an idle device by ~18 % from 0.5 ms to ~20 ms after its start (profiles/r02_experiments.md), and the
    # contract's `--steps 20 --warmup 5` would lie entirely inside that transient.  So about --ramp-seconds of the same
    # updates are queued right AHEAD of the warm-up, with no synchronisation in between: the device is at its steady
    # clocks when the warm-up ends, and the one barrier + synchronise that the contract puts before the timed region is
    # kept as short as possible (the counters read there come from pinned memory, no copy).  Sized from a short pilot.
    ramp_n = 0
    if args.ramp_seconds > 0:
        pilot = np.random.default_rng(args.seed + 4).integers(0, l, size=64).astype(np.uint32)
        eng.run_schedule(pilot[:8])
        eng.synchronize()
        tp = time.perf_counter()
        eng.run_schedule(pilot[8:])
        eng.synchronize()
        per_update = (time.perf_counter() - tp) / 56.0
        if dist is not None:   # every rank must queue the same number of updates
            tpu = torch.tensor([per_update], dtype=torch.float64)
            dist.all_reduce(tpu, op=dist.ReduceOp.MAX)
            per_update = float(tpu.item())
        ramp_n = int(min(20000, max(100, args.ramp_seconds / max(per_update, 1e-6))))
    setup_s = time.time() - t_setup

    locs = np.random.default_rng(args.seed + 3).integers(0, l, size=args.warmup + args.steps).astype(np.uint32)
    ramp_locs = np.random.default_rng(args.seed + 5).integers(0, l, size=ramp_n).astype(np.uint32)

    def device_sync():
        eng.synchronize()                      # the engine's own stream (errors surface here)
        torch.cuda.synchronize(local_rank)     # and the whole device, as the bench contract words it

    def barrier():
        device_sync()
        if dist is not None:
            dist.barrier()

    err, dt, passes, hist = None, 0.0, 0, None
    try:
        barrier()  # (peer-to-peer: a rank's kernels wait at most 3 s for a peer that has not started yet)
        if ramp_n:
            eng.run_schedule(ramp_locs)   # (no synchronisation: the warm-up is queued right behind it)
        if args.warmup:
            eng.run_schedule(locs[:args.warmup])
        barrier()
        p0 = eng.total_passes()
        h0 = eng.pass_histogram()
        t0 = time.perf_counter()
        eng.run_schedule(locs[args.warmup:])
        device_sync()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        passes = eng.total_passes() - p0
        hist = eng.pass_histogram() - h0
    except Exception as exc:  # noqa: BLE001 -- e.g. TSAMD_ECOMM after a peer timed out
        err = exc
    fail_together(err is None, dist, f"timed region: {err}")
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    value = args.steps / dt
    mean_passes = passes / max(1, args.steps)

    m = measure_legs(args, ts, eng, dist, rank, world, local_rank, n, l, k, sc, locs, cores, oracle_ok, gamma_init)
    roofline, cpu, parity, validation, legs = m["roofline"], m["cpu_baseline"], m["parity"], m["validation_block"], m["legs"]

    try:
        recoveries = int(eng.recoveries())
    except Exception:  # noqa: BLE001
        recoveries = None
    if rank == 0:
        alg_update = (mean_passes + 4) * 8.0 * n * k + (mean_passes + 1) * n / 4.0 + 8.0 * n
        out = {
            "metric": f"SNP-minibatch updates/sec (N x K phi+accum) at N={n} K={k}",
            "value": round(value, 2), "unit": "updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"synthetic PSD N={n} individuals x L={l} SNPs, K={k}, 2-bit genotypes "
                                   f"HBM-resident, individuals sharded over {world} GPU(s)",
                       "n": n, "l": l, "k": k, "l_requested": args.l,
                       "parallelism": f"individual-shard x{world}", "exchange": exchange,
                       "exchange_selftest": exchange_report},
            "mean_inner_passes": round(mean_passes, 3),
            "inner_passes_histogram": {str(i): int(c) for i, c in enumerate(hist) if c},
            "nk_pass_per_s": round(value * mean_passes * n * k, 1),
            "update_algorithmic_bytes": alg_update,   # (what the reference's dataflow moves per update)
            "setup_s": round(setup_s, 1),
            "setup": {"clock_ramp_updates": ramp_n,
                      "note": ("untimed priming schedule queued right ahead of the warm-up, no synchronisation in between: the "
                               "device is at its steady clocks when the timed region starts (--ramp-seconds 0 disables)")},
            "roofline": roofline, "cpu_baseline": cpu, "parity_vs_cpu_baseline": parity, "validation_block": validation,
            "legs": legs,
            # resident launches of rank 0's context that found compute units taken and were replayed one launch per pass (0 on a GPU
            # of one's own: a non-zero count means `value` was not measured on the kernel `roofline.kernel` names)
            "recoveries": recoveries,
        }
        if world > 1:
            # what to read an N > 1 line against (stated BEFORE any node has run it: DESIGN.md section 5's predicted table)
            out["predicted_1gpu_equiv"] = ONE_GPU_STEADY.get((n, k))
            out["scaling_note"] = SCALING_NOTE.get(k, SCALING_NOTE[None])
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if parity is not None and not parity["ok"]:
        sys.exit("bench.py: GPU state differs from the CPU baseline's on the same updates: " + json.dumps(parity))


if __name__ == "__main__":
    main()
