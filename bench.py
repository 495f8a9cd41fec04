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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


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
    return ap.parse_args()


def usable_cores():
    """Host cores this process may really use: the smaller of the affinity mask and the
    cgroup CPU quota (a container can see 256 CPUs and be entitled to 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:  # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def choose_exchange(ts, dist, rank, world, local_rank, n, k, theta_shard, gamma_shard):
    """Per-pass exchange of the 2K lambda statistics: direct peer-to-peer stores over xGMI or
    the RCCL all-reduce.  Both are run on this node on the benchmark's own shards (a few SNP
    columns, a short schedule): peer-to-peer is used when it reproduces the RCCL result and is
    not slower.  TSAMD_EXCHANGE=rccl|p2p forces one."""
    import torch

    from terastructure_amd import dist as tdist

    forced = os.environ.get("TSAMD_EXCHANGE", "auto").lower()
    if forced in ("rccl", "p2p"):
        return forced, {}
    l = 32
    beta = np.random.default_rng(7).uniform(0.05, 0.95, size=(l, k))
    locs = np.random.default_rng(8).integers(0, l, size=260).astype(np.uint32)
    locs[:6] = [3, 1, 3, 7, 0, 5]
    out, rate = {}, {}
    for mode in ("rccl", "p2p"):
        out[mode] = None
        e = ts.Engine(n, l, k, device=local_rank, rank=rank, world=world)
        try:
            e.synth_genotypes(theta_shard, beta, seed=11)
            e.set_gamma(gamma_shard)
            try:
                (tdist.bootstrap_p2p if mode == "p2p" else tdist.bootstrap_comm)(e, dist)
            except Exception as exc:  # noqa: BLE001 -- raised on every rank together
                if rank == 0:
                    print(f"[bench] exchange self-test, {mode}: {exc}", file=sys.stderr, flush=True)
                continue
            res, err, dt = None, None, 0.0
            try:
                e.run_schedule(locs[:60])
                e.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                e.run_schedule(locs[60:])
                e.synchronize()
                dt = time.perf_counter() - t0
                res = (e.get_lambda(), e.get_gamma())
            except Exception as exc:  # noqa: BLE001
                err = exc
            if tdist.all_ok(err is None, dist):
                out[mode] = res
                tt = torch.tensor([dt], dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                rate[mode] = round((len(locs) - 60) / float(tt.item()), 1)
            elif rank == 0:
                print(f"[bench] exchange self-test, {mode}: run failed ({err})", file=sys.stderr, flush=True)
        finally:
            dist.barrier()
            e.close()
            dist.barrier()
    ok = out["p2p"] is not None
    if ok and out["rccl"] is not None:
        ok = all(np.allclose(a, b_, rtol=1e-10, atol=0) for a, b_ in zip(out["rccl"], out["p2p"]))
    ok = tdist.all_ok(ok, dist)
    if ok and out["rccl"] is not None and rate["p2p"] < rate["rccl"]:  # (rates are identical on every rank)
        ok = False
    if rank == 0:
        print(f"[bench] exchange self-test: updates/s {rate}, p2p valid {out['p2p'] is not None} -> "
              f"{'p2p' if ok else 'rccl'}", file=sys.stderr, flush=True)
    return ("p2p" if ok else "rccl"), rate


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "TSAMD_BENCH_DEVICE" in os.environ:  # development only: several ranks on one GPU
        local_rank = int(os.environ["TSAMD_BENCH_DEVICE"])
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

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
        import torch

        t = torch.tensor([l], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        l = int(t.item())

    t_setup = time.time()
    # synthetic PSD data: theta ~ Dir(0.2), beta ~ U(0.05, 0.95), y ~ Bin(2, theta.beta)
    rng = np.random.default_rng(args.seed)
    theta = rng.dirichlet(np.full(k, 0.2), size=n)[sb:sb + sc]
    gamma0 = np.random.default_rng(args.seed + 2).gamma(100.0, 0.01, size=(n, k))[sb:sb + sc]
    exchange, exchange_rates = "none", {}
    if world > 1:
        exchange, exchange_rates = choose_exchange(ts, dist, rank, world, local_rank, n, k, theta, gamma0)
    eng = ts.Engine(n, l, k, device=local_rank, rank=rank, world=world)
    if world > 1:
        from terastructure_amd import dist as tdist

        if exchange == "p2p":
            tdist.bootstrap_p2p(eng, dist)
        else:
            tdist.bootstrap_comm(eng, dist)

    chunk = 1 << 17
    brng = np.random.default_rng(args.seed + 1)
    for l0 in range(0, l, chunk):
        beta = brng.uniform(0.05, 0.95, size=(min(chunk, l - l0), k))
        eng.synth_genotypes(theta, beta, first_loc=l0, seed=args.seed)
    eng.set_gamma(gamma0)
    del theta, gamma0
    setup_s = time.time() - t_setup

    locs = np.random.default_rng(args.seed + 3).integers(0, l, size=args.warmup + args.steps).astype(np.uint32)

    import torch

    def device_sync():
        eng.synchronize()                      # the engine's own stream (errors surface here)
        torch.cuda.synchronize(local_rank)     # and the whole device, as the bench contract words it

    def barrier():
        device_sync()
        if dist is not None:
            dist.barrier()

    barrier()  # (peer-to-peer: a rank's kernels wait at most 3 s for a peer that has not started yet)
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
    if dist is not None:
        import torch

        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    value = args.steps / dt
    mean_passes = passes / max(1, args.steps)

    # ---- roofline of the dominant kernel (plain pass, ts_pass<KT,false>) ------------
    # algorithmic bytes per launch = 8*N_shard*K (weights) + N_shard/4 (2-bit column)
    roofline = None
    if not args.no_profile:
        prof_steps = min(args.steps, 300)
        eng.profile_enable(True)
        eng.run_schedule(locs[args.warmup:args.warmup + prof_steps])
        eng.synchronize()
        pr = eng.profile_read()
        eng.profile_enable(False)
        if pr["pass_launches"]:
            avg_s = pr["pass_ms"] / pr["pass_launches"] * 1e-3
            alg_bytes = 8.0 * sc * k + sc / 4.0
            achieved = alg_bytes / avg_s / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pass_kernel_pmc.json")
            if os.path.exists(pmc):
                try:
                    rec = json.load(open(pmc))
                    if rec.get("n") == n and rec.get("k") == k and rec.get("n_gpus") == world:
                        traffic = rec.get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roofline = {
                "bound": "hbm", "kernel": "ts_pass<KT,false>", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_us": round(avg_s * 1e6, 3), "launches_timed": pr["pass_launches"],
                "first_pass_avg_us": round(pr["first_ms"] / max(1, pr["first_launches"]) * 1e3, 3),
            }

    # second denominator: what a plain device-to-device copy reaches on this box,
    # with the benchmark's data still resident (read + write bytes over the copy time)
    if roofline is not None and world == 1:
        try:
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
            roofline["device_copy_GBps"] = round(10 * 2 * src.numel() * 8 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9, 1)
            del src, dst
        except Exception as exc:  # noqa: BLE001 -- an extra, never fatal
            roofline["device_copy_GBps"] = None
            print(f"[bench] device copy probe skipped: {exc}", file=sys.stderr, flush=True)

    # ---- CPU baseline: the oracle ("port") on the host cores, bounded sample ---------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        import oracle_py as op

        cores = args.cpu_threads or usable_cores()
        ls = 8  # sample columns; per-update cost does not depend on L
        orc = op.Oracle(n, ls, k, nthreads=cores, gamma_scale=float(l))
        sample = np.stack([eng.download_bed(int(j)) for j in range(ls)])
        orc.load_bed_payload(sample)
        orc.set_gamma(np.random.default_rng(args.seed + 2).gamma(100.0, 0.01, size=(n, k)))
        done, tc0 = 0, time.perf_counter()
        orc.snp_update(0)  # untimed: first call has no gamma step to apply
        tc0 = time.perf_counter()
        while True:
            orc.snp_update((done + 1) % ls)
            done += 1
            if time.perf_counter() - tc0 > args.cpu_seconds or done >= args.steps:
                break
        cdt = time.perf_counter() - tc0
        cpu = {"value": round(done / cdt, 4), "unit": "SNP-minibatch updates/s", "cores": cores,
               "kind": "port",
               "sample": f"{done} updates (10 passes + gamma step each) at N={n}, K={k} on {ls} of the "
                         f"benchmark's own columns, oracle/ts_oracle.c with {cores} OpenMP threads "
                         f"in the reference's work partition (host shows {os.cpu_count()} CPUs, "
                         f"{cores} usable under its affinity mask / cgroup quota)"}
        orc.close()

    if rank == 0:
        alg_update = (mean_passes + 4) * 8.0 * n * k + (mean_passes + 1) * n / 4.0 + 8.0 * n
        out = {
            "metric": "SNP-minibatch updates/sec (N x K phi+accum) at N=1M K=8",
            "value": round(value, 2), "unit": "updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"synthetic PSD N={n} individuals x L={l} SNPs, K={k}, 2-bit genotypes "
                                   f"HBM-resident, individuals sharded over {world} GPU(s)",
                       "n": n, "l": l, "k": k, "l_requested": args.l,
                       "parallelism": f"individual-shard x{world}", "exchange": exchange,
                       "exchange_selftest_updates_per_s": exchange_rates},
            "mean_inner_passes": round(mean_passes, 3),
            "inner_passes_histogram": {str(i): int(c) for i, c in enumerate(hist) if c},
            "nk_pass_per_s": round(value * mean_passes * n * k, 1),
            "update_algorithmic_bytes": alg_update,
            "update_hbm_frac_of_peak": round(alg_update * value / (world * HBM_PEAK_GBS * 1e9), 4),
            "setup_s": round(setup_s, 1),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
