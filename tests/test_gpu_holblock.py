"""The batched validation block (ts_holblock, csrc/tsamd_holblock_kernels.h).

A validation-mode schedule (compute_likelihood's loop, src/snpsamplinge.cc:476-498: optimize_lambda per validation
location with _hol_mode set, no gamma step in between, src/snpsamplinge.cc:660-668) leaves theta frozen, so the library
runs its locations `batch` at a time: one sweep of the resident weights per sub-batch, ONE in-launch exchange per pass for
the whole batch, per-location convergence.  Every per-location sum keeps the order of the entry-by-entry path, so the
two must agree BIT FOR BIT -- lambda, gamma (the first entry applies the pending step of the last training update), pass
counts, the pass histogram, and everything a training run computes afterwards from the State the block leaves -- and
both agree with the CPU oracle to 1e-9 (c_n and pass counts exact).
"""
import time

import numpy as np
import pytest

import oracle_py as op
from helpers import init_gamma, pack_bed, psd_genotypes, rel_err, slow_params, unpack_bed, usable_cores
from test_gpu_parity import assert_state_close, ts  # noqa: F401

pytestmark = pytest.mark.gpu

TRAIN = np.array([3, 3, 7, 1, 7, 0, 2, 5], dtype=np.uint32)
TRAIN2 = np.array([9, 4, 4, 6, 11, 1], dtype=np.uint32)


def data(n, l, k, seed):
    y, _, _ = psd_genotypes(n, l, k, seed, 0.03)
    return pack_bed(y), init_gamma(n, k, seed + 1), y


def engine(ts, n, l, k, payload, g, y, **cfg):
    eng = ts.Engine(n, l, k, **cfg)
    eng.upload_bed(payload)
    eng.set_gamma(g)
    rng = np.random.default_rng(5)
    held = {}
    for loc in (1, 4, 9, l - 1):
        cand = np.nonzero(y[loc] != 3)[0]
        held[loc] = np.sort(rng.choice(cand, size=max(1, len(cand) // 20), replace=False)).astype(np.uint32)
        eng.set_heldout(loc, held[loc])
    return eng, held


def snapshot(eng):
    return eng.get_lambda(), eng.get_gamma(), eng.get_counts(), eng.total_passes(), eng.pass_histogram()


# (n, k, conv_thresh): K picks the batch (16 locations at K <= 8, 12 at K = 10, 8 at 16, 6 at 20, 4 at 32) and the
# sub-batch (4 / 3 / 2 / 1 locations' accumulators per thread); n the exchange (one workgroup, one level, two levels);
# a raised threshold makes the locations of a batch stop after different pass counts
# (thresholds calibrated with the oracle: mean |dlambda| scales with N and the pass counts flip from 1-2 to 10 within a factor of two)
SHAPES = [(5_000, 6, None), (40_000, 8, None), (40_000, 8, 40_000 / 9000.0), (1_003, 3, None), (200, 3, 0.4), (30_000, 10, 2.5),
          (60_000, 16, None), (45_000, 20, 5.0), (45_000, 20, None), (20_000, 32, None), (3_000, 4, 1.0), (10_000, 6, 2.5), (150_000, 5, None),
          (70_000, 12, 70_000 / 15000.0)]


@pytest.mark.parametrize("n,k,thresh", SHAPES)
def test_block_equals_entry_by_entry_bitwise_and_the_oracle(ts, n, k, thresh, monkeypatch):
    l = 40
    payload, g, y = data(n, l, k, 700 + k)
    over = {} if thresh is None else {"conv_thresh": thresh}
    # a validation list longer than two batches, not a multiple of the batch, ascending like compute_likelihood's map
    vlocs = np.array(sorted(set(range(0, l, 1)) - {3, 17}), dtype=np.uint32)
    outs = []
    for block in (True, False):
        monkeypatch.setenv("TSAMD_HOLBLOCK", "1" if block else "0")
        eng, held = engine(ts, n, l, k, payload, g, y, **over)
        with eng:
            info = eng.holblock_info()
            assert (info["batch"] > 0) == block, info
            snaps = []
            eng.run_schedule(vlocs[:5], 1)           # a block with nothing pending (the run's initial likelihood has no updates, but a caller may)
            eng.run_schedule(TRAIN)
            eng.run_schedule(vlocs, 1)               # the report: first entry applies TRAIN's last gamma step, the rest is batched
            eng.synchronize()
            snaps.append(snapshot(eng))
            eng.run_schedule(TRAIN2)                 # training goes on from the State the block left
            eng.run_schedule(vlocs[::-1], 1)         # any order of distinct locations
            eng.run_schedule(TRAIN2[:2])
            eng.run_schedule(vlocs[:2], 1)           # too short for a block after its first entry: entry by entry
            eng.synchronize()
            snaps.append(snapshot(eng))
            info = eng.holblock_info()
            if block:
                assert info["launches"] == 3 and info["locations"] == 5 + 2 * (len(vlocs) - 1), info
            else:
                assert info["launches"] == 0
            outs.append(snaps)
    for a, b in zip(*outs):
        for x, z in zip(a, b):
            assert np.array_equal(x, z)
    # the oracle on the same sequence
    ocfg = {} if thresh is None else {"meanchangethresh": thresh}
    orc = op.Oracle(n, l, k, nthreads=usable_cores() if n * k > 100_000 else 1, **ocfg)
    orc.load_bed_payload(payload)
    orc.set_gamma(g)
    for loc, ids in held.items():
        orc.set_heldout(loc, ids)
    its = [orc.snp_update(int(x), 1) for x in vlocs[:5]] + [orc.snp_update(int(x)) for x in TRAIN]
    its_block = [orc.snp_update(int(x), 1) for x in vlocs]
    if thresh is not None:
        assert len(set(its_block)) >= 2, its_block      # the locations of a batch really stop at different passes
    its += its_block
    lam, gam, cn, passes, hist = outs[0][0]
    assert passes == sum(its)
    assert all(hist[i] == its.count(i) for i in range(1, 11)), (hist[:12], its)
    assert rel_err(lam, orc.lambda_()) < 1e-9 and rel_err(gam, orc.gamma()) < 1e-9 and np.array_equal(cn, orc.c_indiv())
    its += ([orc.snp_update(int(x)) for x in TRAIN2] + [orc.snp_update(int(x), 1) for x in vlocs[::-1]] + [orc.snp_update(int(x)) for x in TRAIN2[:2]] +
            [orc.snp_update(int(x), 1) for x in vlocs[:2]])
    lam, gam, cn, passes, hist = outs[0][1]
    assert passes == sum(its)
    assert rel_err(lam, orc.lambda_()) < 1e-9 and rel_err(gam, orc.gamma()) < 1e-9 and np.array_equal(cn, orc.c_indiv())
    orc.close()


def test_repeated_locations_cut_the_block(ts, monkeypatch):
    """a location that repeats ends a block (its second visit starts from the first one's lambda)"""
    n, l, k = 20_000, 12, 8
    payload, g, y = data(n, l, k, 91)
    vl = np.array([0, 1, 2, 3, 4, 2, 5, 6, 7, 8, 9, 10, 11, 11, 0, 1, 2], dtype=np.uint32)
    outs = []
    for block in (True, False):
        monkeypatch.setenv("TSAMD_HOLBLOCK", "1" if block else "0")
        eng, _ = engine(ts, n, l, k, payload, g, y)
        with eng:
            eng.run_schedule(TRAIN)
            eng.run_schedule(vl, 1)
            eng.synchronize()
            outs.append(snapshot(eng))
            if block:   # [0] alone through ts_schedule, then {1,2,3,4}, {2,5,...,11}, {11,0,1,2}
                assert eng.holblock_info()["launches"] == 3 and eng.holblock_info()["locations"] == len(vl) - 1
    for x, z in zip(*outs):
        assert np.array_equal(x, z)


def test_heldout_eval_through_the_block_at_size(ts):
    """config 4's shape on one GPU (N = 1M, K = 8: 256 workgroups x 16 individuals per thread, two-level exchange): the report's
    hol-mode updates run as blocks of 16 locations; lambda and the held-out log-likelihood against the oracle."""
    n, l, k = 1_000_000, 40, 8
    rng = np.random.default_rng(8)
    theta = rng.dirichlet(np.full(k, 0.2), size=n)
    beta = rng.uniform(0.05, 0.95, size=(l, k))
    g = rng.gamma(100.0, 0.01, size=(n, k))
    with ts.Engine(n, l, k) as eng:
        eng.synth_genotypes(theta, beta, seed=11, missing_rate=0.01)
        eng.set_gamma(g)
        payload = np.stack([eng.download_bed(j) for j in range(l)])
        vlocs = np.arange(2, 2 + 21, dtype=np.uint32)
        orc = op.Oracle(n, l, k, nthreads=usable_cores())
        orc.load_bed_payload(payload)
        orc.set_gamma(g)
        for loc in vlocs:
            # (like set_validation_sample, src/snpsamplinge.cc:196-224: only observed genotypes are held out)
            cand = np.nonzero(unpack_bed(payload[int(loc)][None, :], n)[0] != 3)[0]
            ids = np.sort(rng.choice(cand, size=2000, replace=False)).astype(np.uint32)
            eng.set_heldout(int(loc), ids)
            orc.set_heldout(int(loc), ids)
        assert eng.holblock_info()["batch"] == 16
        train = np.array([30, 31, 30, 33], dtype=np.uint32)
        eng.run_schedule(train)
        s, c, sums, cnts = eng.heldout_eval(vlocs)
        info = eng.holblock_info()
        assert info["launches"] == 1 and info["locations"] == len(vlocs) - 1, info
        its = [orc.snp_update(int(x)) for x in train] + [orc.snp_update(int(x), 1) for x in vlocs]
        assert eng.total_passes() == sum(its)
        assert_state_close(eng, orc, 1e-9, "block at N = 1M")
        so = [orc.heldout_loglik(int(x)) for x in vlocs]
        assert c == sum(q[1] for q in so)
        assert abs(s - sum(q[0] for q in so)) <= 1e-9 * abs(s)
        orc.close()


def test_block_that_cannot_be_resident_is_replayed(ts, monkeypatch):
    """compute units taken when the block launches: it gives up at its entry exchange, the state intact, and the call is
    replayed one launch per pass from the block's first entry"""
    monkeypatch.setenv("TSAMD_PROBE_MS", "20")
    n, l, k = 300_000, 30, 8
    payload, g, y = data(n, l, k, 57)
    orc = op.Oracle(n, l, k, nthreads=usable_cores())
    orc.load_bed_payload(payload)
    orc.set_gamma(g)
    vlocs = np.arange(0, 25, dtype=np.uint32)
    with ts.Engine(n, l, k) as eng:
        eng.upload_bed(payload)
        eng.set_gamma(g)
        eng.run_schedule(TRAIN)
        eng.run_schedule(vlocs[:1], 1)               # (applies the pending step: the next call is ONE block launch)
        eng.synchronize()
        eng.debug_occupy(160, 400)
        eng.run_schedule(vlocs[1:], 1)               # the block cannot be resident ...
        eng.run_schedule(TRAIN2)                     # ... and what is queued behind it waits for the replay
        passes = eng.total_passes()
        assert eng.recoveries() == 1 and eng.holblock_info()["launches"] == 1
        its = [orc.snp_update(int(x)) for x in TRAIN] + [orc.snp_update(int(x), 1) for x in vlocs] + [orc.snp_update(int(x)) for x in TRAIN2]
        assert passes == sum(its)
        assert_state_close(eng, orc, 1e-9, "after the replay of a block")
        time.sleep(0.5)
        eng.set_launch_mode(ts.LAUNCH_PER_SCHEDULE)
        eng.run_schedule(vlocs, 1)
        eng.synchronize()
        assert eng.recoveries() == 1 and eng.holblock_info()["launches"] == 2
        its += [orc.snp_update(int(x), 1) for x in vlocs]
        assert eng.total_passes() == sum(its)
        assert_state_close(eng, orc, 1e-9, "blocks again after raising the mode")
    orc.close()


@pytest.mark.parametrize("k", slow_params(list(range(1, 33)), [1, 3] + list(range(2, 33, 2)) + [31]))   # (by default the even K, 1, 3 and 31 here and the odd K in tests/test_gpu_hybrid.py; TS_RUN_SLOW=1: every K in both)
def test_every_instantiation_of_the_block(ts, k, monkeypatch):
    """ts_holblock<K> across K = 1 ... 32 (batches of 16 ... 4 locations, sub-batches of 4 / 2 / 1) on a shard of a few
    workgroups: bit for bit the entry-by-entry path, and the oracle at 1e-9"""
    n, l = 3_000 + 97 * k, 24
    payload, g, y = data(n, l, k, 7100 + k)
    vlocs = np.arange(l, dtype=np.uint32)[::-1].copy()
    outs = []
    for block in (True, False):
        monkeypatch.setenv("TSAMD_HOLBLOCK", "1" if block else "0")
        eng, held = engine(ts, n, l, k, payload, g, y)
        with eng:
            eng.run_schedule(TRAIN)
            eng.run_schedule(vlocs, 1)
            eng.run_schedule(TRAIN2[:3])
            eng.synchronize()
            assert (eng.holblock_info()["launches"] == 1) == block
            outs.append(snapshot(eng))
    for x, z in zip(*outs):
        assert np.array_equal(x, z)
    orc = op.Oracle(n, l, k)
    orc.load_bed_payload(payload)
    orc.set_gamma(g)
    for loc, ids in held.items():
        orc.set_heldout(loc, ids)
    its = [orc.snp_update(int(x)) for x in TRAIN] + [orc.snp_update(int(x), 1) for x in vlocs] + [orc.snp_update(int(x)) for x in TRAIN2[:3]]
    lam, gam, cn, passes, _ = outs[0]
    assert passes == sum(its)
    assert rel_err(lam, orc.lambda_()) < 1e-9 and rel_err(gam, orc.gamma()) < 1e-9 and np.array_equal(cn, orc.c_indiv())
    orc.close()
