"""The batched validation block of contexts that run ts_hybrid (ts_hybhol, csrc/tsamd_hybhol_kernels.h).

A shard above ts_schedule's register capacity (N = 1M, K = 20 on one GPU: BASELINE config 5) ran compute_likelihood's loop
(src/snpsamplinge.cc:476-498: optimize_lambda per validation location with _hol_mode set, theta frozen,
src/snpsamplinge.cc:660-668) entry by entry until round 5 -- every location's ten passes re-reading the streamed part of the
weights and paying ten exchanges.  ts_hybhol runs `batch` locations at a time: a sub-batch of them shares one sweep of the
weights (registers + LDS + streamed, split for this kernel), the whole batch shares one exchange per pass.  Every
per-location sum keeps the entry-by-entry path's order, so the two must agree BIT FOR BIT -- lambda, gamma (the first entry
applies the pending step of the last training update), pass counts, the histogram, and the training that follows -- and both
agree with the CPU oracle to 1e-9 (c_n and pass counts exact).
"""
import numpy as np
import pytest

import oracle_py as op
from helpers import init_gamma, pack_bed, psd_genotypes, rel_err, slow_params, usable_cores
from test_gpu_parity import assert_state_close, ts  # noqa: F401

pytestmark = pytest.mark.gpu

TRAIN = np.array([3, 3, 7, 1, 7, 0], dtype=np.uint32)
TRAIN2 = np.array([9, 4, 4, 6], dtype=np.uint32)


def hh_split(k):
    """(locations per exchange, per sweep, register items, LDS items) of ts_hybhol<k> -- mirrors hh_batch / hh_sub /
    hh_reg_items / hh_lds_items in csrc/tsamd_hybhol_kernels.h"""
    ba = 4 if k <= 4 else 2 if k <= 20 else 1
    bx = max(1, min(128 // (k * ba), 16 // ba)) * ba
    budget = (190 if k <= 8 else 140 if k <= 10 else 150 if k <= 13 else 160 if k <= 14 else 175 if k <= 15 else 190 if k <= 16 else 185 if k <= 20
              else 140 if k <= 23 else 165 if k <= 24 else 130 if k <= 27 else 165 if k <= 28 else 100)
    fixed = ba * 2 * k + (ba * 2 * k if k <= 8 else 0) + 2 * k
    reg = max(0, min(16, (budget - fixed) // k))
    j, jx = 2 * k, bx * 2 * k
    batch_lds = bx * 256 * 8 + bx * 4 * j * 8 + 2 * bx * j * 8 + max(jx, 4 * j) * 8 + bx * 4 + 1024
    lds = min((160 * 1024 - batch_lds) // (k * 8 * 256), 16 - reg)
    return bx, ba, reg, lds


def snapshot(eng):
    return eng.get_lambda(), eng.get_gamma(), eng.get_counts(), eng.total_passes(), eng.pass_histogram()


def run_both(ts, monkeypatch, n, l, k, seed, thresh=None, flags=0):
    """(snapshots with the block, snapshots entry by entry, the oracle's state after the same entries, held-out sets)"""
    y, _, _ = psd_genotypes(n, l, k, seed, 0.02)
    payload, g = pack_bed(y), init_gamma(n, k, seed + 1)
    rng = np.random.default_rng(seed + 2)
    held = {}
    for loc in (1, 4, l - 1):
        cand = np.nonzero(y[loc] != 3)[0]
        held[loc] = np.sort(rng.choice(cand, size=max(1, n // 100), replace=False)).astype(np.uint32)
    del y
    vlocs = np.array(sorted(set(range(l)) - {3}), dtype=np.uint32)
    over = {} if thresh is None else {"conv_thresh": thresh}
    outs = []
    for block in (True, False):
        monkeypatch.setenv("TSAMD_HOLBLOCK", "1" if block else "0")
        with ts.Engine(n, l, k, flags=flags, **over) as eng:
            eng.upload_bed(payload)
            eng.set_gamma(g)
            for loc, ids in held.items():
                eng.set_heldout(loc, ids)
            geo = eng.schedule_geometry()
            assert geo["on_chip_per_thread"] <= geo["indivs_per_thread"] and eng.launch_info()["kernels_per_snp"] == 0
            info = eng.holblock_info()
            assert info["batch"] == (hh_split(k)[0] if block else 0), info
            snaps = []
            eng.run_schedule(TRAIN)
            eng.run_schedule(vlocs, 1)               # the report: first entry through ts_hybrid (pending gamma step), the rest batched
            eng.synchronize()
            snaps.append(snapshot(eng))
            eng.run_schedule(TRAIN2)                 # training goes on from the State the block left
            eng.run_schedule(vlocs[::-1][:5], 1)     # a block shorter than a batch, any order
            eng.run_schedule(TRAIN2[:2])
            eng.synchronize()
            snaps.append(snapshot(eng))
            info = eng.holblock_info()
            assert info["launches"] == (2 if block else 0) and info["locations"] == ((len(vlocs) - 1) + 4 if block else 0), info
            outs.append(snaps)
    orc = op.Oracle(n, l, k, nthreads=usable_cores() if n * k > 100_000 else 1, **({} if thresh is None else {"meanchangethresh": thresh}))
    orc.load_bed_payload(payload)
    orc.set_gamma(g)
    for loc, ids in held.items():
        orc.set_heldout(loc, ids)
    its = [orc.snp_update(int(x)) for x in TRAIN]
    its_block = [orc.snp_update(int(x), 1) for x in vlocs]
    return outs, orc, its, its_block


def check(outs, orc, its, its_block, thresh):
    for a, b in zip(*outs):                          # the block and the entry-by-entry path: the same bits everywhere
        for x, z in zip(a, b):
            assert np.array_equal(x, z)
    if thresh is not None:
        assert len(set(its_block)) >= 2, its_block   # the locations of a batch really stop at different passes
    lam, gam, cn, passes, hist = outs[0][0]
    allits = its + its_block
    assert passes == sum(allits) and all(hist[i] == allits.count(i) for i in range(1, 11)), (hist[:12], allits)
    assert rel_err(lam, orc.lambda_()) < 1e-9 and rel_err(gam, orc.gamma()) < 1e-9 and np.array_equal(cn, orc.c_indiv())
    orc.close()


# (n, k, conv_thresh): above ts_schedule's capacity; K picks the batch and the split, n the streamed items per thread
@pytest.mark.parametrize("n,k,thresh", [(1_000_000, 20, 15.0), (2_000_000, 8, None), (1_200_000, 12, 30.0), (300_000, 32, None), (1_100_000, 3, None)])
def test_hybrid_block_equals_entry_by_entry_bitwise_and_the_oracle(ts, n, k, thresh, monkeypatch):
    outs, orc, its, its_block = run_both(ts, monkeypatch, n, 16, k, 9100 + k, thresh)
    check(outs, orc, its, its_block, thresh)


@pytest.mark.parametrize("k", slow_params(list(range(1, 33)), [1, 4, 5, 8, 9, 12, 13, 14, 16, 17, 20, 21, 24, 27, 29, 32]))
def test_instantiations_of_the_hybrid_block_on_a_small_device(ts, k, monkeypatch):
    """ts_hybhol<K> across K (every split: 4 / 2 / 1 locations per sweep, register / LDS / streamed items) on the launch
    geometry of a device with four compute units (TSAMD_TEST_MAX_WORKGROUPS, honoured with TSAMD_FLAG_TEST_HOOKS only), so that
    a few thousand individuals fill every item class: bit for bit the entry-by-entry path, and the oracle at 1e-9"""
    monkeypatch.setenv("TSAMD_TEST_MAX_WORKGROUPS", "4")
    reg = 16 if k <= 8 else 128 // k if k <= 16 else 112 // k if k <= 24 else 3
    if k > 20:
        reg -= 1
    chip = reg + min(16, (160 * 1024 - 1024 - 200 * k) // (k * 8 * 256))   # ts_hybrid's on-chip items: the context must be above ts_schedule's
    n = 4 * 256 * (chip + 3) - 37                                           # capacity; three items beyond ts_hybrid's own on-chip share
    outs, orc, its, its_block = run_both(ts, monkeypatch, n, 14, k, 9300 + k, None, flags=ts.FLAG_TEST_HOOKS)
    check(outs, orc, its, its_block, None)


def test_hybrid_block_that_cannot_be_resident_is_replayed(ts, monkeypatch):
    """a tenant holds compute units when the block is launched: its entry exchange gives up with the state intact, the
    schedule is replayed one launch per pass (tsamd_recoveries) and the results are the oracle's"""
    import time

    n, l, k = 600_000, 12, 20
    y, _, _ = psd_genotypes(n, l, k, 77, 0.02)
    payload, g = pack_bed(y), init_gamma(n, k, 78)
    del y
    orc = op.Oracle(n, l, k, nthreads=usable_cores())
    orc.load_bed_payload(payload)
    orc.set_gamma(g)
    vlocs = np.arange(l, dtype=np.uint32)
    monkeypatch.setenv("TSAMD_PROBE_MS", "30")
    with ts.Engine(n, l, k) as eng:
        eng.upload_bed(payload)
        eng.set_gamma(g)
        eng.run_schedule(TRAIN)
        eng.synchronize()
        eng.debug_occupy(200, 400)
        time.sleep(0.2)
        eng.run_schedule(vlocs, 1)
        eng.synchronize()
        assert eng.recoveries() == 1, eng.last_error()
        its = [orc.snp_update(int(x)) for x in TRAIN] + [orc.snp_update(int(x), 1) for x in vlocs]
        assert eng.total_passes() == sum(its)
        assert_state_close(eng, orc, 1e-9, "replayed hybrid block")
        # ... and with NO gamma step pending (advisor, round 5): above, a training update precedes the report, so the launch that
        # fails at its entry is the one-entry ts_hybrid that applies the pending step, and the ts_hybhol launches queued behind it
        # are merely aborted.  Here the previous call was a validation block: the first launch of the next block IS a ts_hybhol --
        # its own entry failure, found in the journal by its launch offset, and its replay.
        time.sleep(0.5)
        eng.set_launch_mode(ts.LAUNCH_PER_SCHEDULE)
        eng.run_schedule(vlocs[:7], 1)                  # batched again (ts_hybhol), undisturbed
        eng.synchronize()
        blocks = eng.holblock_info()["launches"]
        assert blocks >= 1 and eng.recoveries() == 1
        eng.debug_occupy(200, 400)
        time.sleep(0.2)
        eng.run_schedule(vlocs[::-1], 1)               # no step pending: every launch of this call is a ts_hybhol; the first gives up
        eng.run_schedule(TRAIN2)
        eng.synchronize()
        assert eng.recoveries() == 2, eng.last_error()
        assert eng.holblock_info()["launches"] > blocks   # (the failing launch WAS a block launch)
        its += [orc.snp_update(int(x), 1) for x in vlocs[:7]] + [orc.snp_update(int(x), 1) for x in vlocs[::-1]] + [orc.snp_update(int(x)) for x in TRAIN2]
        assert eng.total_passes() == sum(its)
        assert_state_close(eng, orc, 1e-9, "replayed ts_hybhol launch")
    orc.close()
