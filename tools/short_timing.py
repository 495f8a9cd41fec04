"""Where a 20-update timed region's time goes on the host side (run on the GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import terastructure_amd as ts
n, l, k = 1_000_000, 4000, 8
rng = np.random.default_rng(1)
e = ts.Engine(n, l, k)
e.synth_genotypes(rng.dirichlet(np.full(k, 0.2), size=n), rng.uniform(0.05, 0.95, size=(l, k)), seed=3)
e.set_gamma(rng.gamma(100, 0.01, size=(n, k)))
e.prepare()
locs = rng.integers(0, l, size=4000).astype(np.uint32)
e.run_schedule(locs[:5]); e.synchronize()
for steps in (20, 20, 20, 100, 1000):
    torch.cuda.synchronize(0)
    t0 = time.perf_counter()
    e.run_schedule(locs[:steps])
    t1 = time.perf_counter()
    e.synchronize()
    t2 = time.perf_counter()
    torch.cuda.synchronize(0)
    t3 = time.perf_counter()
    print(f"steps {steps}: submit {1e6*(t1-t0):.0f} us, wait {1e6*(t2-t1):.0f} us, torch sync {1e6*(t3-t2):.0f} us, per step {1e6*(t3-t0)/steps:.2f} us")
