"""Soak run on the GPU box (not part of the test suite): a long schedule, many short schedules and single updates
through ts_schedule at N = 1M, K = 8; checks that nothing times out, that the state stays finite and that every gamma
row sum keeps the invariant of SURVEY section 4 (S -> (1 - rho) S + rho (K alpha + 2 L) per step: between the initial
sum and K alpha + 2 L).   usage: python tools/soak.py [updates in the long schedule]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import terastructure_amd as ts

n, l, k = 1_000_000, 50_000, 8
long_n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rng = np.random.default_rng(5)
e = ts.Engine(n, l, k)
theta = rng.dirichlet(np.full(k, 0.2), size=n)
for l0 in range(0, l, 1 << 14):
    e.synth_genotypes(theta, rng.uniform(0.05, 0.95, size=(min(1 << 14, l - l0), k)), first_loc=l0, seed=3)
g0 = rng.gamma(100, 0.01, size=(n, k))
e.set_gamma(g0)
print("mode", e.launch_info(), flush=True)
t0 = time.time()
e.run_schedule(rng.integers(0, l, size=long_n).astype(np.uint32))
e.synchronize()
t1 = time.time()
print(f"long schedule: {long_n} updates in {t1 - t0:.1f} s = {long_n / (t1 - t0):.0f} updates/s, passes {e.total_passes()}", flush=True)
for i in range(3000):
    e.run_schedule(rng.integers(0, l, size=int(rng.integers(1, 8))).astype(np.uint32), hol_mode=int(i % 17 == 0))
e.synchronize()
t2 = time.time()
its = [e.snp_update(int(x)) for x in rng.integers(0, l, size=3000)]
t3 = time.time()
print(f"3000 short schedules {t2 - t1:.1f} s, 3000 single updates {t3 - t2:.1f} s ({3000 / (t3 - t2):.0f} /s), passes per update {np.mean(its):.2f}", flush=True)
gam, lam = e.get_gamma(), e.get_lambda()
assert np.all(np.isfinite(gam)) and np.all(np.isfinite(lam)) and gam.min() > 0 and lam.min() > 0
s = gam.sum(axis=1)
target = k * e.cfg.alpha + 2.0 * l
lo, hi = np.minimum(g0.sum(axis=1), target), np.maximum(g0.sum(axis=1), target)
assert np.all(s >= lo * (1 - 1e-9)) and np.all(s <= hi * (1 + 1e-9)), (s.min(), s.max(), target)
print(f"ok: gamma row sums in [{s.min():.1f}, {s.max():.1f}], target {target:.1f}; updates by passes run {dict((i, int(v)) for i, v in enumerate(e.pass_histogram()) if v)}")
