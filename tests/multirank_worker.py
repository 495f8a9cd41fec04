"""One rank of the sharded engine (launched by test_gpu_multirank.py, RANK / WORLD_SIZE /
MASTER_* in the environment).  All ranks may share one GPU (TS_DEVICE): the peer-to-peer
exchange only needs IPC-mapped buffers, which also works between processes on one device."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    import terastructure_amd as ts
    from terastructure_amd import dist as tdist
    from helpers import init_gamma, pack_bed, psd_genotypes

    out_dir, mode = sys.argv[1], sys.argv[2]
    n, l, k, seed, nsnp = (int(x) for x in sys.argv[3:8])
    d, rank, world = tdist.init_process_group("gloo")
    device = int(os.environ.get("TS_DEVICE", rank))
    y, _, _ = psd_genotypes(n, l, k, seed, 0.03)
    payload = pack_bed(y)
    gamma = init_gamma(n, k, seed + 1)
    over = {}
    if os.environ.get("TS_CONV_THRESH"):      # early convergence: SNPs stop after differing pass counts
        over["conv_thresh"] = float(os.environ["TS_CONV_THRESH"])
    if os.environ.get("TS_MAX_INNER"):
        over["max_inner"] = int(os.environ["TS_MAX_INNER"])
    if os.environ.get("TS_DELAY_RANK") == str(rank):  # this rank stalls between its flag wait and its row reads
        os.environ["TSAMD_TEST_XCHG_DELAY_US"] = os.environ.get("TS_DELAY_US", "200")
    flags = int(os.environ.get("TS_FLAGS", "0"))
    if any(v.startswith("TSAMD_TEST_") for v in os.environ):
        flags |= ts.FLAG_TEST_HOOKS           # the library honours its test hooks only when asked to
    eng = ts.Engine(n, l, k, device=device, rank=rank, world=world, flags=flags, **over)
    b, c = eng.shard_begin, eng.shard_count
    eng.upload_bed(payload)
    eng.set_gamma(gamma[b:b + c])
    rng = np.random.default_rng(seed + 2)
    for loc in rng.choice(l, size=max(1, l // 8), replace=False):
        cand = np.nonzero(y[loc] != 3)[0]
        eng.set_heldout(int(loc), rng.choice(cand, size=max(1, n // 50), replace=False))
    if mode == "p2p":
        tdist.bootstrap_p2p(eng, d)   # (TSAMD_SCHEDULE_GATHER=leaders in the environment: three-level cross-rank exchange)
    else:
        err = None
        try:
            tdist.bootstrap_comm(eng, d)
        except Exception as exc:  # noqa: BLE001 -- e.g. RCCL refuses two ranks on one device
            err = exc
        if not tdist.all_ok(err is None, d):
            print(f"RCCL communicator unavailable: {err}", flush=True)
            d.barrier()
            eng.close()
            d.destroy_process_group()
            sys.exit(77)
    if os.environ.get("TS_LAUNCH_MODE"):      # e.g. 0: one launch per pass although the shard qualifies for ts_schedule
        eng.set_launch_mode(int(os.environ["TS_LAUNCH_MODE"]))
    kps = eng.launch_info()["kernels_per_snp"]
    if os.environ.get("TS_EXPECT_KPS"):       # which kernel sequence the test means to exercise
        assert kps == int(os.environ["TS_EXPECT_KPS"]), f"kernels per SNP: {kps}"
    if os.environ.get("TS_EXPECT_HYBRID"):    # the shard exceeds ts_schedule's register capacity: ts_hybrid (more individuals per thread than its register items)
        geo = eng.schedule_geometry()
        reg_items = 16 if k <= 8 else 128 // k if k <= 16 else 4 if k == 22 else 112 // k if k <= 24 else 3
        assert geo["indivs_per_thread"] > reg_items, geo
    if os.environ.get("TS_EXPECT_PER_THREAD"):  # the instantiation a case means to run (16 at K <= 8: the one without the skip-unused-items branches)
        geo = eng.schedule_geometry()
        assert geo["indivs_per_thread"] == int(os.environ["TS_EXPECT_PER_THREAD"]), geo
    locs = np.random.default_rng(seed + 3).integers(0, l, size=nsnp).astype(np.uint32)
    switch = os.environ.get("TS_SWITCH_MODES") == "1"   # ts_schedule -> one launch per pass -> ts_schedule, mid-run
    eng.run_schedule(locs[:5])          # eager path
    eng.synchronize()
    if switch:
        eng.set_launch_mode(ts.LAUNCH_PER_PASS)
    its = [eng.snp_update(int(locs[5]))]
    if switch:
        eng.run_schedule(locs[6:20])
        eng.set_launch_mode(ts.LAUNCH_PER_SCHEDULE)
        eng.run_schedule(locs[20:])
    else:
        if os.environ.get("TS_OCCUPY") and rank == 0:   # "workgroups,milliseconds": a tenant takes compute units of rank 0's device
            wgs, ms = (int(x) for x in os.environ["TS_OCCUPY"].split(","))
            eng.debug_occupy(wgs, ms)
            import time
            time.sleep(0.3)             # (until ALL its workgroups hold their compute units, not just the first)
        if os.environ.get("TS_OCCUPY"):
            d.barrier()
        eng.run_schedule(locs[6:])      # graph replay path when long enough
    try:
        eng.synchronize()
    except Exception:
        print(f"rank {rank}: synchronize failed; recoveries so far {eng.recoveries()}", flush=True)
        raise
    if os.environ.get("TS_HOL_LOCS"):         # a validation block (distinct locations, hol mode), then training again
        nhol = int(os.environ["TS_HOL_LOCS"])
        eng.run_schedule(np.arange(nhol, dtype=np.uint32), 1)
        eng.run_schedule(locs[:4])
        eng.synchronize()
        info = eng.holblock_info()
        want = int(os.environ.get("TS_EXPECT_HOLBLOCKS", "-1"))
        assert want < 0 or info["launches"] == want, (info, eng.recoveries(), eng.last_error(), eng.launch_info())
    if os.environ.get("TS_EXPECT_RECOVERIES"):
        assert eng.recoveries() == int(os.environ["TS_EXPECT_RECOVERIES"]), f"recoveries: {eng.recoveries()} ({eng.last_error()})"
        if int(os.environ["TS_EXPECT_RECOVERIES"]) > 0:
            assert eng.launch_info()["kernels_per_snp"] == eng.cfg.max_inner
    full = tdist.gather_rows(eng.get_gamma(), n, d, ts.shard_range)
    cnt = tdist.gather_rows(eng.get_counts().astype(np.float64)[:, None], n, d, ts.shard_range)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), lam=eng.get_lambda(), gamma=full, cnt=cnt, its=np.array(its),
             passes=eng.total_passes(), kps=kps, recoveries=eng.recoveries())
    d.barrier()
    eng.close()
    d.barrier()
    d.destroy_process_group()


if __name__ == "__main__":
    main()
