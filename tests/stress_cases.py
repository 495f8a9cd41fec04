"""Randomised device-vs-oracle parity cases, shared by tools/stress_parity.py (hundreds of cases, by hand) and
tests/test_gpu_geometry.py (a fixed-seed slice in the suite): (n, k, missing rate, pass cap, launch mode, cut into
calls) drawn at random; inner pass counts and c_n exactly, lambda / gamma to 1e-9.  K = 1 ... 40 covers the resident
kernels' every geometry (16 ... 3 individuals per thread, rows exchanged over 1 ... 4 waves, one workgroup / one level /
two levels) and the run-time-K fallback."""
import numpy as np

import oracle_py as op
from helpers import init_gamma, pack_bed, psd_genotypes, rel_err, usable_cores

SIZES = [1, 7, 64, 200, 513, 1000, 2999, 5000, 12000, 70001, 150000, 150000, 260000]


def run_case(ts, rng, sizes=SIZES):
    """one random case; returns (ok, description)"""
    n = int(rng.choice(sizes))
    k = int(rng.integers(1, 41))
    if n > 200000:
        k = min(k, 24)   # (the oracle's time per case)
    l = 12
    miss = float(rng.choice([0.0, 0.02, 0.3]))
    cap = int(rng.choice([1, 3, 10, 10, 10, 40]))
    seed = int(rng.integers(1 << 30))
    y, _, _ = psd_genotypes(n, l, k, seed, miss)
    payload = pack_bed(y)
    g = init_gamma(n, k, seed + 1)
    if rng.random() < 0.2:
        g = g * 10.0 ** rng.uniform(-2, 3, size=g.shape)
    eng = ts.Engine(n, l, k, max_inner=cap)
    orc = op.Oracle(n, l, k, online_iterations=cap, nthreads=usable_cores() if n > 20000 else 1)
    try:
        mode = int(rng.integers(0, 4))   # 3: whatever tsamd_create chose
        if mode < 3:
            try:
                eng.set_launch_mode(mode)
            except ts.TsamdError:
                mode = 3
        eng.upload_bed(payload)
        orc.load_bed_payload(payload)
        eng.set_gamma(g)
        orc.set_gamma(g)
        for loc in rng.choice(l, size=2, replace=False):
            cand = np.nonzero(y[loc] != 3)[0]
            if len(cand):
                ids = np.sort(rng.choice(cand, size=max(1, len(cand) // 10), replace=False)).astype(np.uint32)
                eng.set_heldout(int(loc), ids)
                orc.set_heldout(int(loc), ids)
        locs = rng.integers(0, l, size=24).astype(np.uint32)
        hol = rng.random(24) < 0.1
        its_o = [orc.snp_update(int(a), int(h)) for a, h in zip(locs, hol)]
        if rng.random() < 0.5:
            its_d = [eng.snp_update(int(a), int(h)) for a, h in zip(locs, hol)]
        else:  # schedules (graph replay when long enough), split at the hol entries
            its_d = None
            i = 0
            while i < len(locs):
                j = i
                while j < len(locs) and hol[j] == hol[i]:
                    j += 1
                eng.run_schedule(locs[i:j], hol_mode=int(hol[i]))
                i = j
            eng.synchronize()
        ok = (its_d is None or its_d == its_o) and eng.total_passes() == sum(its_o)
        el, eg = rel_err(eng.get_lambda(), orc.lambda_()), rel_err(eng.get_gamma(), orc.gamma())
        ok = ok and el < 1e-9 and eg < 1e-9 and np.array_equal(eng.get_counts(), orc.c_indiv())
        desc = (f"n={n} k={k} miss={miss} cap={cap} mode={mode} seed={seed} passes {eng.total_passes()} vs {sum(its_o)} "
                f"lambda {el:.2e} gamma {eg:.2e}")
    finally:
        eng.close()
        orc.close()
    return ok, desc
