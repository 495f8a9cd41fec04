"""Shards above the register capacity of ts_schedule (ts_hybrid, csrc/tsamd_hybrid_kernels.h).

The reference takes any -n / -k (src/main.cc:115-123).  ts_schedule holds 256 x 256 x resident_items(K) individuals' weights
in registers (1 048 576 at K <= 8, 327 680 at K = 20); one more and, until round 4, the context fell back to ten launches per
update.  ts_hybrid keeps the one-launch structure and splits a thread's individuals between registers, LDS (weights, for
the whole launch) and memory (re-read every pass; their gamma step streams and writes back the weights as well).  Against
the CPU oracle (rel 1e-9 on lambda / gamma, c_n and pass counts exact) for every item class: LDS items only, LDS + streamed
items, an odd and an even number of streamed items (the pipeline takes two per turn), thresholds that stop SNPs early,
validation-mode entries, any cut of the schedule into calls, mode switches in the middle of a run, and a launch that cannot
be co-resident.  BASELINE config 5 on ONE GPU (N = 1M, K = 20) and its 2-GPU shard (500K) run here.
"""
import time

import numpy as np
import pytest

import oracle_py as op
from helpers import init_gamma, pack_bed, psd_genotypes, rel_err, slow_params, usable_cores
from test_gpu_parity import assert_state_close, ts  # noqa: F401

pytestmark = pytest.mark.gpu

LOCS = np.array([3, 3, 7, 1, 7, 7, 0, 2, 2, 5, 9, 11, 4, 4, 6, 8], dtype=np.uint32)

# (n, k) -> (workgroups, individuals per thread, of which on chip)
# (round 5: (400K, 20), (600K, 20) and (1.2M, 12) left this list -- the suite's time budget; those classes -- LDS items partly
# filled, an even and an odd number of streamed items -- run in test_hybrid_with_early_stops_and_any_cut, tests/test_gpu_hybhol.py
# and the sharded tests of tests/test_gpu_multirank.py)
SHAPES = {(500_000, 20): (255, 8, 8),        # config 5's 2-GPU shard: exactly registers + LDS
          (1_000_000, 20): (256, 16, 8),     # config 5 on one GPU: eight streamed items
          (2_000_000, 8): (256, 31, 25),     # six streamed items
          (1_100_000, 8): (256, 17, 17),     # one LDS item
          (300_000, 32): (254, 5, 3),        # two streamed items; ONE register item (round 6: two spilled 236 bytes to scratch)
          (1_100_000, 3): (256, 17, 17)}
# (all workgroups take an equal share of the shard -- a multiple of 16 individuals -- so a workgroup's last 256-thread round
# is partly filled: the kernel is bound by memory and 245 of 256 compute units would leave bandwidth unused)


def pair(ts, n, l, k, seed, thresh=None):
    y, _, _ = psd_genotypes(n, l, k, seed, 0.02)
    payload = pack_bed(y)
    g = init_gamma(n, k, seed + 1)
    rng = np.random.default_rng(seed + 2)
    held = {}
    for loc in (2, 7):
        cand = np.nonzero(y[loc] != 3)[0]
        held[loc] = np.sort(rng.choice(cand, size=n // 100, replace=False)).astype(np.uint32)
    del y
    orc = op.Oracle(n, l, k, nthreads=usable_cores(), **({} if thresh is None else {"meanchangethresh": thresh}))
    orc.load_bed_payload(payload)
    orc.set_gamma(g)
    eng = ts.Engine(n, l, k, **({} if thresh is None else {"conv_thresh": thresh}))
    eng.upload_bed(payload)
    eng.set_gamma(g)
    for loc, ids in held.items():
        eng.set_heldout(loc, ids)
        orc.set_heldout(loc, ids)
    return eng, orc


@pytest.mark.parametrize("n,k", sorted(SHAPES))
def test_hybrid_matches_the_oracle(ts, n, k):
    l = 12
    eng, orc = pair(ts, n, l, k, 8000 + k)
    with eng:
        geo = eng.schedule_geometry()
        assert (geo["workgroups"], geo["indivs_per_thread"], geo["on_chip_per_thread"]) == SHAPES[(n, k)], geo
        assert eng.launch_info()["kernels_per_snp"] == 0 and eng.holblock_info()["batch"] > 0   # (validation-mode schedules: ts_hybhol, tests/test_gpu_hybhol.py)
        eng.run_schedule(LOCS[:9])
        eng.run_schedule(LOCS[9:11], 1)      # validation-mode updates: no gamma step follows them
        eng.run_schedule(LOCS[11:12])
        eng.run_schedule(LOCS[12:])
        eng.synchronize()
        its = [orc.snp_update(int(x), 1 if 9 <= i < 11 else 0) for i, x in enumerate(LOCS)]
        assert eng.total_passes() == sum(its)
        assert_state_close(eng, orc, 1e-9, f"ts_hybrid n {n} k {k}")
        # single updates (a launch per call: the weights go out to memory and come back), then the other mode and back
        for x in (5, 5, 1):
            assert eng.snp_update(int(x)) == orc.snp_update(int(x))
        eng.set_launch_mode(ts.LAUNCH_PER_PASS)
        assert eng.launch_info()["kernels_per_snp"] == 10
        eng.run_schedule(LOCS[:4])
        eng.set_launch_mode(ts.LAUNCH_PER_SCHEDULE)
        eng.run_schedule(LOCS[4:8])
        eng.synchronize()
        for x in LOCS[:8]:
            orc.snp_update(int(x))
        assert_state_close(eng, orc, 1e-9, f"ts_hybrid n {n} k {k} after mode switches")
    orc.close()


@pytest.mark.parametrize("n,k,thresh", [(600_000, 20, 15.0), (1_200_000, 12, 30.0), (2_000_000, 8, 50.0)])
def test_hybrid_with_early_stops_and_any_cut(ts, n, k, thresh):
    """SNPs that stop after 1 ... 10 passes; the same schedule in one call and cut into calls gives the same bits"""
    l = 12
    eng, orc = pair(ts, n, l, k, 8100 + k, thresh)
    outs = []
    with eng:
        eng.run_schedule(LOCS)
        eng.synchronize()
        its = [orc.snp_update(int(x)) for x in LOCS]
        assert len(set(its)) >= 2, its
        assert eng.total_passes() == sum(its)
        hist = eng.pass_histogram()
        assert all(hist[i] == its.count(i) for i in range(1, 11)), (hist[:12], its)
        assert_state_close(eng, orc, 1e-9, f"ts_hybrid early stops n {n} k {k}")
        outs.append((eng.get_lambda(), eng.get_gamma(), eng.get_counts()))
    orc.close()
    eng2, orc2 = pair(ts, n, l, k, 8100 + k, thresh)
    orc2.close()
    with eng2:
        for a, b in ((0, 1), (1, 7), (7, 8), (8, 19), (19, len(LOCS))):
            eng2.run_schedule(LOCS[a:b])
        eng2.synchronize()
        outs.append((eng2.get_lambda(), eng2.get_gamma(), eng2.get_counts()))
    for x, z in zip(*outs):
        assert np.array_equal(x, z)


def test_hybrid_launch_that_cannot_be_resident_is_replayed(ts, monkeypatch):
    monkeypatch.setenv("TSAMD_PROBE_MS", "20")
    n, l, k = 600_000, 12, 20
    eng, orc = pair(ts, n, l, k, 8200)
    with eng:
        eng.run_schedule(LOCS[:5])
        eng.synchronize()
        eng.debug_occupy(160, 400)
        eng.run_schedule(LOCS[5:11])                 # cannot be resident: gives up at its entry exchange, the state intact
        eng.run_schedule(LOCS[11:14], 1)
        passes = eng.total_passes()                  # the replay, one launch per pass
        assert eng.recoveries() == 1 and eng.launch_info()["kernels_per_snp"] == 10
        its = [orc.snp_update(int(x), 1 if 11 <= i < 14 else 0) for i, x in enumerate(LOCS[:14])]
        assert passes == sum(its)
        assert_state_close(eng, orc, 1e-9, "ts_hybrid after the replay")
        time.sleep(0.5)
        eng.set_launch_mode(ts.LAUNCH_PER_SCHEDULE)
        eng.run_schedule(LOCS[14:])
        eng.synchronize()
        its += [orc.snp_update(int(x)) for x in LOCS[14:]]
        assert eng.total_passes() == sum(its) and eng.recoveries() == 1
        assert_state_close(eng, orc, 1e-9, "ts_hybrid raised again")
    orc.close()


def _on_chip_items(k):
    """(register items, register + LDS items) of ts_hybrid<K>: hy_reg_items / hy_lds_items, csrc/tsamd_hybrid_kernels.h"""
    reg = 16 if k <= 8 else 13 if k == 9 else 128 // k if k <= 16 else 112 // k if k <= 20 else 112 // k - 1 if k <= 24 else 2 if k <= 28 else 1
    return reg, reg + min(16, (160 * 1024 - 1024 - 200 * k) // (k * 8 * 256))


@pytest.mark.parametrize("k", slow_params(list(range(1, 33)), [2] + list(range(1, 33, 2)) + [32]))   # (by default the odd K, 2 and 32 here and the even K in tests/test_gpu_holblock.py: the suite's time budget; TS_RUN_SLOW=1: every K in both)
def test_every_instantiation_on_a_small_device(ts, k, monkeypatch):
    """ts_hybrid<K> across K = 1 ... 32 -- with and without streamed items -- on four workgroups (TSAMD_TEST_MAX_WORKGROUPS:
    the geometry of a device with four compute units), so that a few thousand individuals fill the register items, the LDS
    items and three streamed items of every thread; against the oracle, with validation-mode entries and a repeated location."""
    monkeypatch.setenv("TSAMD_TEST_MAX_WORKGROUPS", "4")
    reg, chip = _on_chip_items(k)
    l = 10
    locs = np.array([3, 3, 7, 1, 7, 0, 2, 5, 9, 4, 4, 6], dtype=np.uint32)
    for extra in (3, -1):          # three streamed items per thread / one item short of the on-chip capacity (no streaming)
        n = 4 * 256 * (chip + extra) - 37
        if n <= 4 * 256 * reg:     # (K where LDS adds a single item: "one short" is ts_schedule's territory)
            continue
        y, _, _ = psd_genotypes(n, l, k, 8300 + k, 0.03)
        payload = pack_bed(y)
        g = init_gamma(n, k, 8301 + k)
        orc = op.Oracle(n, l, k, nthreads=usable_cores() if n * k > 100_000 else 1)
        orc.load_bed_payload(payload)
        orc.set_gamma(g)
        with ts.Engine(n, l, k, flags=ts.FLAG_TEST_HOOKS) as eng:
            eng.upload_bed(payload)
            eng.set_gamma(g)
            geo = eng.schedule_geometry()
            assert geo["workgroups"] == 4 and geo["on_chip_per_thread"] == min(chip, geo["indivs_per_thread"]), geo
            assert geo["indivs_per_thread"] == chip + extra and geo["indivs_per_thread"] > reg, geo
            eng.run_schedule(locs[:7])
            eng.run_schedule(locs[7:9], 1)
            eng.run_schedule(locs[9:])
            eng.synchronize()
            its = [orc.snp_update(int(x), 1 if 7 <= i < 9 else 0) for i, x in enumerate(locs)]
            assert eng.total_passes() == sum(its)
            assert_state_close(eng, orc, 1e-9, f"ts_hybrid<{k}> on four workgroups, {chip + extra} individuals per thread")
        orc.close()
