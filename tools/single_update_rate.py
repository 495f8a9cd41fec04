#!/usr/bin/env python3
"""tsamd_snp_update (one call = one update + a synchronise) in each launch mode: which one should a caller that cannot
batch be routed to?   usage: python3 tools/single_update_rate.py [n] [k] [calls]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import terastructure_amd as ts  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 600
l = 4096
rng = np.random.default_rng(1)
theta = rng.dirichlet(np.full(k, 0.2), size=n)
with ts.Engine(n, l, k) as eng:
    eng.synth_genotypes(theta, rng.uniform(0.05, 0.95, size=(l, k)), seed=3)
    eng.set_gamma(rng.gamma(100.0, 0.01, size=(n, k)))
    locs = rng.integers(0, l, size=calls + 50)
    for name, mode in (("per schedule", ts.LAUNCH_PER_SCHEDULE), ("per SNP", ts.LAUNCH_PER_SNP), ("per pass", ts.LAUNCH_PER_PASS),
                       ("per schedule", ts.LAUNCH_PER_SCHEDULE)):
        try:
            eng.set_launch_mode(mode)
        except ts.TsamdError as exc:
            print(f"{name}: {exc}")
            continue
        eng.prepare()
        for x in locs[:50]:
            eng.snp_update(int(x))
        t0 = time.perf_counter()
        for x in locs[50:]:
            eng.snp_update(int(x))
        dt = time.perf_counter() - t0
        print(f"N={n} K={k} tsamd_snp_update, launch {name}: {calls / dt:.0f} calls/s ({dt / calls * 1e6:.1f} us per call)", flush=True)
