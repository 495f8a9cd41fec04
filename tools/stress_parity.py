"""Randomised parity stress (not part of the test suite): many (n, k, missing rate, pass cap, launch mode)
combinations, device vs oracle: inner pass counts exactly, lambda / gamma to 1e-9.  K = 1 ... 40 covers the resident
kernels' every geometry (16 ... 3 individuals per thread, rows exchanged over 1 ... 4 waves) and the run-time-K fallback.
python tools/stress_parity.py [cases] [seed]"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import terastructure_amd as ts
import oracle_py as op
from helpers import init_gamma, pack_bed, psd_genotypes, rel_err

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    n = int(rng.choice([1, 7, 64, 200, 513, 1000, 2999, 5000, 12000, 70001, 150000, 150000, 260000]))
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
    eng = ts.Engine(n, l, k, max_inner=cap); orc = op.Oracle(n, l, k, online_iterations=cap, nthreads=16 if n > 20000 else 1)
    mode = int(rng.integers(0, 4))   # 3: whatever tsamd_create chose
    if mode < 3:
        try:
            eng.set_launch_mode(mode)
        except ts.TsamdError:
            mode = 3
    eng.upload_bed(payload); orc.load_bed_payload(payload); eng.set_gamma(g); orc.set_gamma(g)
    for loc in rng.choice(l, size=2, replace=False):
        cand = np.nonzero(y[loc] != 3)[0]
        if len(cand):
            ids = np.sort(rng.choice(cand, size=max(1, len(cand) // 10), replace=False)).astype(np.uint32)
            eng.set_heldout(int(loc), ids); orc.set_heldout(int(loc), ids)
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
            eng.run_schedule(locs[i:j], hol_mode=int(hol[i])); i = j
        eng.synchronize()
    ok = (its_d is None or its_d == its_o) and eng.total_passes() == sum(its_o)
    el, eg = rel_err(eng.get_lambda(), orc.lambda_()), rel_err(eng.get_gamma(), orc.gamma())
    ok = ok and el < 1e-9 and eg < 1e-9 and np.array_equal(eng.get_counts(), orc.c_indiv())
    if not ok:
        bad += 1
        print(f"MISMATCH case {c}: n={n} k={k} miss={miss} cap={cap} mode={mode} seed={seed} passes {eng.total_passes()} vs {sum(its_o)} lambda {el:.2e} gamma {eg:.2e}", flush=True)
    eng.close(); orc.close()
print(f"{cases} cases, {bad} mismatches")
