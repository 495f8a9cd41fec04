"""The three ways the library launches the same arithmetic (tsamd_set_launch_mode): one kernel per pass,
first pass + one resident kernel per SNP (ts_resident), one kernel per schedule (ts_schedule: the weights
stay in registers from the first SNP to the last, the gamma step reads and writes gamma only).  Every mode
is compared with the CPU oracle (rel 1e-9, pass counts and c_n exact), the modes with each other (they
differ by the order in which the workgroups' partial rows are added: rel 1e-11), and each mode must give
the same BITS however the schedule is cut into calls and whichever mode ran before it.  K = 1 ... 32: the
resident kernels hold pairs of individuals per item at K <= 8 and single individuals above, exchange rows of
2K values over one, two or four waves, and (ts_schedule) defer the exchange of a SNP's last pass into the
next SNP's first one unless one of the next two SNPs revisits the location -- LOCS has every such pattern.
"""
import numpy as np
import pytest

import oracle_py as op
from helpers import init_gamma, pack_bed, psd_genotypes, rel_err, usable_cores
from test_gpu_parity import assert_state_close, ts  # noqa: F401

pytestmark = pytest.mark.gpu

LOCS = np.array([3, 3, 7, 1, 7, 7, 7, 0, 2, 2, 5, 9, 11, 4, 4, 6, 8, 10, 3, 1, 0, 0, 5, 5, 2], dtype=np.uint32)


def make(ts, n, l, k, seed, **cfg):
    y, _, _ = psd_genotypes(n, l, k, seed, 0.03)
    payload = pack_bed(y)
    g = init_gamma(n, k, seed + 1)
    eng = ts.Engine(n, l, k, **cfg)
    eng.upload_bed(payload)
    eng.set_gamma(g)
    ocfg = {"online_iterations": cfg["max_inner"]} if "max_inner" in cfg else {}
    orc = op.Oracle(n, l, k, nthreads=usable_cores() if n * k > 100_000 else 1, **ocfg)
    orc.load_bed_payload(payload)
    orc.set_gamma(g)
    return eng, orc, payload, g


def state(eng):
    return eng.get_lambda(), eng.get_gamma(), eng.get_counts(), eng.total_passes()


@pytest.mark.parametrize("n,k,max_inner", [(1003, 4, 10), (40_000, 8, 10), (40_000, 5, 3), (70_000, 8, 1), (200_000, 3, 10),
                                           (64, 1, 10), (5, 2, 10), (40_000, 8, 2), (30_000, 9, 10), (50_000, 12, 10),
                                           (60_000, 16, 10), (45_000, 20, 10), (20_000, 32, 10), (9_000, 17, 3), (70_000, 24, 2),
                                           (33_000, 25, 10), (700, 13, 10), (150_000, 10, 10), (80_000, 14, 10)])
def test_every_mode_matches_the_oracle_and_the_others(ts, n, k, max_inner):
    l = 12
    outs = {}
    for mode in (ts.LAUNCH_PER_PASS, ts.LAUNCH_PER_SNP, ts.LAUNCH_PER_SCHEDULE):
        eng, orc, _, _ = make(ts, n, l, k, 900 + n % 97, max_inner=max_inner)
        with eng:
            if mode == ts.LAUNCH_PER_SNP and max_inner < 2:
                with pytest.raises(ts.TsamdError):
                    eng.set_launch_mode(mode)   # (no plain passes to keep resident)
                continue
            if mode == ts.LAUNCH_PER_SCHEDULE and max_inner < 2:
                with pytest.raises(ts.TsamdError):
                    eng.set_launch_mode(mode)
                continue
            eng.set_launch_mode(mode)
            want = {ts.LAUNCH_PER_PASS: max_inner, ts.LAUNCH_PER_SNP: 2, ts.LAUNCH_PER_SCHEDULE: 0}[mode]
            assert eng.launch_info()["kernels_per_snp"] == want
            eng.run_schedule(LOCS[:9])
            eng.run_schedule(LOCS[9:10], 1)      # one validation-mode update: no gamma step follows it
            eng.run_schedule(LOCS[10:])
            eng.synchronize()
            its = [orc.snp_update(int(x), 1 if i == 9 else 0) for i, x in enumerate(LOCS)]
            assert eng.total_passes() == sum(its)
            assert_state_close(eng, orc, 1e-9, f"mode {mode}")
            outs[mode] = state(eng)
    base = outs[ts.LAUNCH_PER_PASS]
    for mode, got in outs.items():
        assert rel_err(got[0], base[0]) < 1e-11 and rel_err(got[1], base[1]) < 1e-11, mode
        assert np.array_equal(got[2], base[2]) and got[3] == base[3], mode


@pytest.mark.parametrize("mode_name", ["LAUNCH_PER_SNP", "LAUNCH_PER_SCHEDULE"])
@pytest.mark.parametrize("n,k", [(40_000, 8), (1003, 3)])
def test_cuts_and_single_updates_give_the_same_bits(ts, mode_name, n, k):
    """one call / three calls / one tsamd_snp_update per entry"""
    l = 12
    mode = getattr(ts, mode_name)
    outs = []
    for cut in ("whole", "pieces", "single"):
        eng, _, _, _ = make(ts, n, l, k, 77)
        with eng:
            eng.set_launch_mode(mode)
            eng.prepare()
            if cut == "whole":
                eng.run_schedule(LOCS)
            elif cut == "pieces":
                eng.run_schedule(LOCS[:1])
                eng.run_schedule(LOCS[1:14])
                eng.synchronize()
                eng.run_schedule(LOCS[14:])
            else:
                for x in LOCS:
                    eng.snp_update(int(x))
            eng.synchronize()
            outs.append(state(eng))
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert np.array_equal(a, b)


def test_switching_modes_in_the_middle_of_a_run(ts):
    """per schedule -> per pass -> per SNP -> per schedule, validation-mode updates in between: the State one
    mode leaves (last SNP complete, gamma step pending) is the next one's start"""
    n, l, k = 50_000, 12, 6
    eng, orc, _, _ = make(ts, n, l, k, 31)
    with eng:
        seq = [(ts.LAUNCH_PER_SCHEDULE, LOCS[:6], 0), (ts.LAUNCH_PER_PASS, LOCS[6:11], 0), (ts.LAUNCH_PER_SNP, LOCS[11:13], 1),
               (ts.LAUNCH_PER_SCHEDULE, LOCS[13:19], 0), (ts.LAUNCH_PER_SNP, LOCS[19:22], 0), (ts.LAUNCH_PER_SCHEDULE, LOCS[22:], 0)]
        for mode, locs, hol in seq:
            eng.set_launch_mode(mode)
            eng.run_schedule(locs, hol)
            for x in locs:
                orc.snp_update(int(x), hol)
        eng.synchronize()
        assert_state_close(eng, orc, 1e-9, "mode switches")


def test_modes_a_context_does_not_qualify_for(ts, monkeypatch):
    with ts.Engine(2000, 4, 40) as eng:            # K above 32: the run-time-K fallback kernels, one launch per pass
        assert eng.launch_info()["kernels_per_snp"] == eng.cfg.max_inner
        for mode in (ts.LAUNCH_PER_SNP, ts.LAUNCH_PER_SCHEDULE):
            with pytest.raises(ts.TsamdError):
                eng.set_launch_mode(mode)
        eng.set_launch_mode(ts.LAUNCH_PER_PASS)
    with ts.Engine(400_000, 4, 20) as eng:         # K = 20 holds 5 individuals per thread in registers: 327 680 per GPU ...
        assert eng.launch_info()["kernels_per_snp"] == 0   # ... above that the whole-schedule kernel is ts_hybrid
        geo = eng.schedule_geometry()
        assert geo["indivs_per_thread"] == 7 and geo["on_chip_per_thread"] == 7 and geo["workgroups"] == 256, geo
        with pytest.raises(ts.TsamdError):
            eng.set_launch_mode(ts.LAUNCH_PER_SNP)         # (ts_resident has no such variant)
    monkeypatch.setenv("TSAMD_HYBRID", "0")
    with ts.Engine(400_000, 4, 20) as eng:         # without it: one launch per pass
        assert eng.launch_info()["kernels_per_snp"] == eng.cfg.max_inner
        with pytest.raises(ts.TsamdError):
            eng.set_launch_mode(ts.LAUNCH_PER_SCHEDULE)
    monkeypatch.delenv("TSAMD_HYBRID")
    with ts.Engine(327_680, 4, 20) as eng:
        assert eng.launch_info()["kernels_per_snp"] == 0
    with ts.Engine(2000, 4, 4, nodekappa=0.7) as eng:   # the whole-schedule kernel has the reference's default exponent built in
        assert eng.launch_info()["kernels_per_snp"] == 2
        with pytest.raises(ts.TsamdError):
            eng.set_launch_mode(ts.LAUNCH_PER_SCHEDULE)
        with pytest.raises(ts.TsamdError):
            eng.set_launch_mode(7)


@pytest.mark.parametrize("n", [30_000, 6_000])      # 118 workgroups: two-level exchange; 24: one level (every workgroup reads every row, 16 row pairs per lane)
@pytest.mark.parametrize("k", [8, 12, 20])
def test_deferred_last_exchange_patterns(ts, k, n):
    """ts_schedule parks the row of a SNP's last pass (under the pass cap) for the next SNP's first exchange unless the
    next or the next-but-one SNP is at the same location, the launch ends, or the SNP stopped early.  Every such
    neighbourhood, with a pass cap of 2 (the deferred pass directly follows the first) and of 10, with converging SNPs
    mixed in, cut into launches at every position: against the oracle and bitwise against uncut."""
    l = 6
    locs = np.array([0, 1, 2, 0, 0, 3, 4, 3, 5, 5, 5, 1, 2, 1, 0, 4, 4, 2, 3, 3, 1, 5, 0, 2], dtype=np.uint32)
    for max_inner, thresh in ((10, None), (2, None), (10, 6.0), (3, 2.0)):
        over = {} if thresh is None else {"conv_thresh": thresh}
        eng, orc, payload, g = make(ts, n, l, k, 400 + k, max_inner=max_inner, **over)
        if thresh is not None:
            orc.close()
            orc = op.Oracle(n, l, k, online_iterations=max_inner, meanchangethresh=thresh, nthreads=usable_cores())
            orc.load_bed_payload(payload)
            orc.set_gamma(g)
        with eng:
            assert eng.launch_info()["kernels_per_snp"] == 0
            eng.run_schedule(locs)
            eng.synchronize()
            its = [orc.snp_update(int(x)) for x in locs]
            if thresh is not None:
                assert len(set(its)) >= 2, its
            assert eng.total_passes() == sum(its)
            assert_state_close(eng, orc, 1e-9, f"max_inner {max_inner} thresh {thresh}")
            whole = state(eng)
        for cut in (1, 2, 5, 11, 23):
            eng2 = ts.Engine(n, l, k, max_inner=max_inner, **over)
            with eng2:
                eng2.upload_bed(payload)
                eng2.set_gamma(g)
                eng2.run_schedule(locs[:cut])
                eng2.run_schedule(locs[cut:])
                eng2.synchronize()
                for a, b in zip(state(eng2), whole):
                    assert np.array_equal(a, b), (max_inner, thresh, cut)


def test_other_learning_rate_exponent_runs_per_snp_and_matches(ts):
    n, l, k = 3000, 8, 4
    y, _, _ = psd_genotypes(n, l, k, 5, 0.02)
    payload = pack_bed(y)
    g = init_gamma(n, k, 6)
    eng = ts.Engine(n, l, k, nodekappa=0.7)
    orc = op.Oracle(n, l, k, nodekappa=0.7)
    eng.upload_bed(payload)
    orc.load_bed_payload(payload)
    eng.set_gamma(g)
    orc.set_gamma(g)
    with eng:
        for x in LOCS[:10] % l:
            assert eng.snp_update(int(x)) == orc.snp_update(int(x))
        assert_state_close(eng, orc, 1e-9, "nodekappa 0.7")
