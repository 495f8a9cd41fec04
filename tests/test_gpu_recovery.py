"""A resident launch that cannot get all its workgroups onto the device at once must not cost the run.

ts_schedule / ts_resident exchange partial sums between their workgroups inside the launch, which needs all of them
resident together.  Every such launch checks that with its first exchange, before it modifies anything; when another
tenant holds compute units (tsamd_debug_occupy plays one) the launch gives up with the state intact, every kernel
queued behind it becomes a no-op, and the next synchronising call lowers the context to one launch per pass, replays
the affected schedules from the unchanged state and reports success with a warning.  The results are those of an
undisturbed run: against the oracle (rel 1e-9, c_n and pass counts exact).  (The reference's counterpart: SIGTERM
saves the model instead of losing it, src/snpsamplinge.cc:454-457.)
"""
import time

import numpy as np
import pytest

import oracle_py as op
from helpers import init_gamma, pack_bed, psd_genotypes, usable_cores
from test_gpu_parity import assert_state_close, ts  # noqa: F401

pytestmark = pytest.mark.gpu

LOCS = np.array([3, 3, 7, 1, 7, 7, 7, 0, 2, 2, 5, 9, 11, 4, 4, 6, 8, 10, 3, 1, 0, 0, 5, 5, 2, 6, 9, 9, 1], dtype=np.uint32)


@pytest.mark.parametrize("mode_name,n,k", [("LAUNCH_PER_SCHEDULE", 300_000, 8), ("LAUNCH_PER_SNP", 300_000, 8),
                                           ("LAUNCH_PER_SCHEDULE", 200_000, 16), ("LAUNCH_PER_SNP", 150_000, 20)])
def test_launch_that_cannot_be_resident_is_replayed(ts, mode_name, n, k, monkeypatch):
    monkeypatch.setenv("TSAMD_PROBE_MS", "20")
    l = 12
    mode = getattr(ts, mode_name)
    y, _, _ = psd_genotypes(n, l, k, 55, 0.02)
    payload = pack_bed(y)
    g = init_gamma(n, k, 56)
    del y
    orc = op.Oracle(n, l, k, nthreads=usable_cores())
    orc.load_bed_payload(payload)
    orc.set_gamma(g)
    with ts.Engine(n, l, k) as eng:
        eng.upload_bed(payload)
        eng.set_gamma(g)
        eng.set_launch_mode(mode)
        kps = eng.launch_info()["kernels_per_snp"]
        eng.run_schedule(LOCS[:6])                   # undisturbed
        eng.synchronize()
        assert eng.recoveries() == 0
        eng.debug_occupy(160, 400)                   # another tenant: 160 of the 256 compute units for 0.4 s
        eng.run_schedule(LOCS[6:13])                 # this launch cannot be resident ...
        eng.run_schedule(LOCS[13:16], 1)             # ... and everything queued behind it waits for the replay
        eng.run_schedule(LOCS[16:22])
        passes = eng.total_passes()                  # (a synchronising call: the replay happens here)
        assert eng.recoveries() == 1
        assert "warning" in eng.last_error() and "replayed" in eng.last_error()
        assert eng.launch_info()["kernels_per_snp"] == eng.cfg.max_inner   # lowered to one launch per pass
        its = [orc.snp_update(int(x), 1 if 13 <= i < 16 else 0) for i, x in enumerate(LOCS[:22])]
        assert passes == sum(its)
        assert_state_close(eng, orc, 1e-9, "after the replay")
        time.sleep(0.5)                              # the other tenant leaves
        eng.set_launch_mode(mode)                    # raised again: resident launches work as before
        assert eng.launch_info()["kernels_per_snp"] == kps
        eng.run_schedule(LOCS[22:])
        eng.synchronize()
        assert eng.recoveries() == 1
        its += [orc.snp_update(int(x)) for x in LOCS[22:]]
        assert eng.total_passes() == sum(its)
        assert_state_close(eng, orc, 1e-9, "after raising the mode again")
    orc.close()


def test_single_updates_survive_a_tenant(ts, monkeypatch):
    """tsamd_snp_update (one launch per call) while compute units are held: the first call is replayed, the later ones
    run one launch per pass; pass counts come back right every time."""
    monkeypatch.setenv("TSAMD_PROBE_MS", "10")
    n, l, k = 260_000, 8, 6
    y, _, _ = psd_genotypes(n, l, k, 8, 0.02)
    payload = pack_bed(y)
    g = init_gamma(n, k, 9)
    del y
    orc = op.Oracle(n, l, k, nthreads=usable_cores())
    orc.load_bed_payload(payload)
    orc.set_gamma(g)
    with ts.Engine(n, l, k) as eng:
        eng.upload_bed(payload)
        eng.set_gamma(g)
        assert eng.snp_update(2) == orc.snp_update(2)
        eng.debug_occupy(200, 300)
        for x in (5, 5, 1, 0):
            assert eng.snp_update(x) == orc.snp_update(x)
        assert eng.recoveries() == 1
        assert_state_close(eng, orc, 1e-9, "single updates")
    orc.close()


def test_two_engines_sharing_one_gpu(ts, monkeypatch):
    """Two independent contexts on the same GPU, each large enough to want most of its compute units for a whole-schedule
    launch (196 workgroups each, 256 compute units): their launches overlap in time, so at least one of them finds the
    device taken at its entry exchange.  Neither run may fail or be poisoned: the affected context replays one launch per
    pass; both end in the state they reach when run one after the other (the undisturbed one bit for bit, a replayed one
    to the rounding between launch modes)."""
    monkeypatch.setenv("TSAMD_PROBE_MS", "15")
    n, l, k, nupd = 300_000, 16, 8, 1500
    rng = np.random.default_rng(77)
    data = []
    for s in (1, 2):
        y, _, _ = psd_genotypes(n, l, k, 90 + s, 0.02)
        data.append((pack_bed(y), init_gamma(n, k, 95 + s), rng.integers(0, l, size=nupd).astype(np.uint32)))
        del y

    def run(concurrent):
        engs = [ts.Engine(n, l, k) for _ in data]
        try:
            for e, (payload, g, _) in zip(engs, data):
                e.upload_bed(payload)
                e.set_gamma(g)
            if concurrent:
                for e, (_, _, locs) in zip(engs, data):
                    e.run_schedule(locs)            # asynchronous: the second launch starts while the first one runs
                for e in engs:
                    e.synchronize()
            else:
                for e, (_, _, locs) in zip(engs, data):
                    e.run_schedule(locs)
                    e.synchronize()
            return [(e.get_lambda(), e.get_gamma(), e.get_counts(), e.total_passes(), e.recoveries()) for e in engs]
        finally:
            for e in engs:
                e.close()

    alone = run(False)
    assert [a[4] for a in alone] == [0, 0]
    together = run(True)
    assert sum(t[4] for t in together) >= 1, "the two launches never overlapped (nothing was tested)"
    for a, t in zip(alone, together):
        assert np.array_equal(a[2], t[2]) and a[3] == t[3]
        if t[4] == 0:
            assert np.array_equal(a[0], t[0]) and np.array_equal(a[1], t[1])
        else:
            from helpers import rel_err

            # 1 500 updates carry the launch modes' different summation orders forward: 1e-11 in lambda, 1e-8 in gamma
            assert rel_err(t[0], a[0]) < 1e-9 and rel_err(t[1], a[1]) < 1e-6
