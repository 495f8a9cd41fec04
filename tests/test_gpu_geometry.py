"""The launch geometries of the resident kernels that small and mid-sized shards take (resident_geometry in
csrc/tsamd.hip), each against the CPU oracle in every launch mode the context qualifies for:

  * the SHRUNK grid: a shard that would fill 33 ... 80 workgroups with one individual per thread is launched on <= 32
    workgroups with two or three per thread, so that its in-launch exchange has ONE level; at K <= 8 every wave then runs
    the K x 2 epilogue for itself (kRepl).  BASELINE config 2 (N = 10 000, K = 6) runs exactly there: 20 workgroups x 2;
  * ONE workgroup with several individuals per thread: nothing is exchanged at all.

tsamd_schedule_geometry reports what the context chose; the tests assert that the intended branch was taken before they
compare (rel 1e-9 on lambda / gamma, c_n and pass counts exact, with and without a raised threshold so that SNPs stop after
1 ... 10 passes).  Plus a fixed-seed slice of the randomised stress of tools/stress_parity.py.
Reference: PhiRunnerE::process / update_lambda_t (src/snpsamplinge.hh:416-431, src/snpsamplinge.cc:742-759) -- the sums
these geometries split differently."""
import numpy as np
import pytest

import oracle_py as op
from helpers import init_gamma, pack_bed, psd_genotypes, rel_err, usable_cores
from test_gpu_parity import assert_state_close, ts  # noqa: F401

pytestmark = pytest.mark.gpu

LOCS = np.array([3, 3, 7, 1, 7, 7, 0, 2, 2, 5, 9, 11, 4, 4, 6, 8, 10, 3, 1, 0], dtype=np.uint32)

# (n, k) -> (workgroups, individuals per thread, exchange levels) of ts_schedule
SHRUNK = {(10_000, 6): (20, 2, 1), (10_000, 8): (20, 2, 1), (16_000, 8): (32, 2, 1), (20_000, 3): (27, 3, 1)}
ONE_WG = {(1_500, 8): (1, 6, 0), (3_000, 4): (1, 12, 0)}
# mean |dlambda| scales with N: a threshold of N / this stops the SNPs of LOCS after 1 ... 10 passes (calibrated with the oracle)
THRESH_DIV = {(10_000, 6): 4000.0, (10_000, 8): 4000.0, (16_000, 8): 6000.0, (20_000, 3): 4000.0, (1_500, 8): 2500.0, (3_000, 4): 2500.0}


def pair(ts, n, l, k, seed, thresh=None):
    y, _, _ = psd_genotypes(n, l, k, seed, 0.03)
    payload = pack_bed(y)
    g = init_gamma(n, k, seed + 1)
    over = {} if thresh is None else {"conv_thresh": thresh}
    eng = ts.Engine(n, l, k, **over)
    ocfg = {} if thresh is None else {"meanchangethresh": thresh}
    orc = op.Oracle(n, l, k, nthreads=usable_cores() if n * k > 100_000 else 1, **ocfg)
    eng.upload_bed(payload)
    orc.load_bed_payload(payload)
    eng.set_gamma(g)
    orc.set_gamma(g)
    rng = np.random.default_rng(seed + 2)
    for loc in (2, 7):
        cand = np.nonzero(y[loc] != 3)[0]
        ids = np.sort(rng.choice(cand, size=max(1, len(cand) // 20), replace=False)).astype(np.uint32)
        eng.set_heldout(loc, ids)
        orc.set_heldout(loc, ids)
    return eng, orc


@pytest.mark.parametrize("thresh", [None, "raised"])
@pytest.mark.parametrize("n,k", sorted(SHRUNK) + sorted(ONE_WG))
def test_small_shard_geometries_match_the_oracle(ts, n, k, thresh):
    l = 12
    want = SHRUNK.get((n, k)) or ONE_WG[(n, k)]
    th = None if thresh is None else n / THRESH_DIV[(n, k)]
    for mode in (ts.LAUNCH_PER_SCHEDULE, ts.LAUNCH_PER_SNP, ts.LAUNCH_PER_PASS):
        eng, orc = pair(ts, n, l, k, 3000 + n % 101 + k, th)
        with eng:
            geo = eng.schedule_geometry(ts.LAUNCH_PER_SCHEDULE)
            assert (geo["workgroups"], geo["indivs_per_thread"], geo["exchange_levels"]) == want, geo
            geo1 = eng.schedule_geometry(ts.LAUNCH_PER_SNP)   # ts_resident: one level only up to 16 workgroups
            assert geo1["exchange_levels"] == (0 if want[0] == 1 else 1 if geo1["workgroups"] <= 16 else 2), geo1
            eng.set_launch_mode(mode)
            assert eng.launch_info()["kernels_per_snp"] == {ts.LAUNCH_PER_SCHEDULE: 0, ts.LAUNCH_PER_SNP: 2, ts.LAUNCH_PER_PASS: 10}[mode]
            eng.run_schedule(LOCS[:8])
            eng.run_schedule(LOCS[8:9], 1)      # one validation-mode update: no gamma step follows it
            eng.run_schedule(LOCS[9:])
            eng.synchronize()
            its = [orc.snp_update(int(x), 1 if i == 8 else 0) for i, x in enumerate(LOCS)]
            if th is not None:
                assert len(set(its)) >= 2, its    # SNPs really stop at different pass counts
            assert eng.total_passes() == sum(its), (mode, its)
            hist = eng.pass_histogram()
            assert all(hist[i] == its.count(i) for i in range(1, 11)), (mode, hist[:12], its)
            assert_state_close(eng, orc, 1e-9, f"n {n} k {k} mode {mode} thresh {th}")
            # ... and through single updates (a launch per call: loads and writes back the weights around ONE update)
            for x in (5, 5, 1):
                assert eng.snp_update(int(x)) == orc.snp_update(int(x))
            assert_state_close(eng, orc, 1e-9, f"n {n} k {k} mode {mode} single updates")
        orc.close()


def test_geometry_query_at_other_sizes(ts):
    """full-size and mid-size shards: 256 workgroups, two levels; a context that does not qualify says so"""
    with ts.Engine(1_048_576, 2, 8) as eng:
        assert eng.schedule_geometry() == dict(workgroups=256, indivs_per_thread=16, exchange_levels=2, on_chip_per_thread=16)
    with ts.Engine(100_000, 2, 8) as eng:
        geo = eng.schedule_geometry()
        assert geo["indivs_per_thread"] == 2 and geo["exchange_levels"] == 2 and 190 <= geo["workgroups"] <= 200, geo
    with ts.Engine(200, 2, 3) as eng:      # config 1's shape: one workgroup
        assert eng.schedule_geometry() == dict(workgroups=1, indivs_per_thread=2, exchange_levels=0, on_chip_per_thread=2)
    with ts.Engine(2000, 4, 40) as eng:    # K above 32: no resident kernel
        with pytest.raises(ts.TsamdError):
            eng.schedule_geometry()
        with pytest.raises(ts.TsamdError):
            eng.schedule_geometry(ts.LAUNCH_PER_PASS)


def test_random_geometries_slice(ts):
    """8 fixed-seed cases of the randomised stress (tests/stress_cases.py) -- 40 with TS_RUN_SLOW=1, as until round 5 (the suite's
    time budget); the tool runs hundreds: tools/stress_parity.py"""
    import os

    from stress_cases import run_case

    rng = np.random.default_rng(20240)
    bad = []
    for c in range(40 if os.environ.get("TS_RUN_SLOW", "0") not in ("", "0") else 8):
        ok, desc = run_case(ts, rng)
        if not ok:
            bad.append((c, desc))
    assert not bad, bad


def test_workgroup_cap_knob_changes_the_geometry_not_the_results(ts, monkeypatch):
    """TSAMD_SCHED_WORKGROUPS (round 6's tuning knob: fewer, fatter workgroups for the resident kernels -- the geometry sweep
    of profiles/r06_experiments.md): the capped context runs more individuals per thread on fewer workgroups and reaches the
    oracle's state all the same; the uncapped one is the default geometry."""
    n, l, k = 60_000, 12, 8
    eng, orc = pair(ts, n, l, k, 3300)
    with eng:
        base = eng.schedule_geometry()
        eng.run_schedule(LOCS)
        eng.synchronize()
        its = [orc.snp_update(int(x)) for x in LOCS]
        assert eng.total_passes() == sum(its)
        assert_state_close(eng, orc, 1e-9, "default geometry")
        lam0, gam0 = eng.get_lambda(), eng.get_gamma()
    orc.close()
    monkeypatch.setenv("TSAMD_SCHED_WORKGROUPS", "64")
    eng, orc = pair(ts, n, l, k, 3300)
    with eng:
        geo = eng.schedule_geometry()
        assert base["workgroups"] > 64 and geo["workgroups"] <= 64 and geo["indivs_per_thread"] > base["indivs_per_thread"], (base, geo)
        eng.run_schedule(LOCS)
        eng.synchronize()
        assert rel_err(eng.get_lambda(), lam0) < 1e-10 and rel_err(eng.get_gamma(), gam0) < 1e-10   # (another order of the partial rows)
    orc.close()
