"""The validation block (compute_likelihood, src/snpsamplinge.cc:461-544) at BASELINE config 4 on one GPU: the reference's
validation sample (0.5 % of the locations, N/100 held-out individuals each: 5 000 x 10 000 entries at N = L = 1M) and
the time of one report: a hol-mode schedule over the validation locations + one evaluation kernel
(tsamd_heldout_eval; batched through ts_holblock, TSAMD_HOLBLOCK=0 for the entry-by-entry path).
K = 20 (N = 1M: a context that runs ts_hybrid) batches through ts_hybhol.
usage (GPU box): python tools/validation_block.py [L = 1000000] [K = 8]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import terastructure_amd as ts

n = 1_000_000
l = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(3)
e = ts.Engine(n, l, k)
theta = rng.dirichlet(np.full(k, 0.2), size=n)
for l0 in range(0, l, 1 << 17):
    e.synth_genotypes(theta, rng.uniform(0.05, 0.95, size=(min(1 << 17, l - l0), k)), first_loc=l0, seed=3)
e.set_gamma(rng.gamma(100, 0.01, size=(n, k)))
vlocs = np.sort(rng.choice(l, size=max(1, l // 200), replace=False)).astype(np.uint32)
t0 = time.time()
for loc in vlocs:
    e.set_heldout(int(loc), np.sort(rng.choice(n, size=n // 100, replace=False)).astype(np.uint32))
t1 = time.time()
print(f"validation sample: {len(vlocs)} locations x {n // 100} individuals, set_heldout {t1 - t0:.1f} s", flush=True)
e.run_schedule(rng.integers(0, l, size=2000).astype(np.uint32))      # some training first
e.synchronize()
for rep in range(3):
    e.run_schedule(rng.integers(0, l, size=500).astype(np.uint32))
    e.synchronize()
    t0 = time.time()
    s, c, _, _ = e.heldout_eval(vlocs, run_updates=True)
    t1 = time.time()
    t2 = time.time()
    e.heldout_eval(vlocs, run_updates=False)
    t3 = time.time()
    print(f"report {rep}: {len(vlocs)} hol-mode updates + evaluation of {c} entries in {t1 - t0:.3f} s "
          f"({(t1 - t0) / len(vlocs) * 1e6:.1f} us per validation location; the evaluation alone: {t3 - t2:.3f} s), "
          f"mean log-likelihood {s / c:.6f}; {e.holblock_info()}", flush=True)
