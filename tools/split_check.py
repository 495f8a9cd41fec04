"""single-GPU sequence vs sharded sequence (one shard) at a size with several row batches"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import terastructure_amd as ts
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
l, k = 64, 8
rng = np.random.default_rng(1)
theta = rng.dirichlet(np.full(k, 0.2), size=n); beta = rng.uniform(0.05, 0.95, size=(l, k))
gamma = rng.gamma(100.0, 0.01, size=(n, k)); locs = rng.integers(0, l, size=int(sys.argv[2]) if len(sys.argv) > 2 else 3).astype(np.uint32)
out = {}
for mode in ("single", "split", "split-nograph"):
    fl = 0 if mode == "single" else ts.FLAG_SPLIT_EPILOGUE
    if mode.endswith("nograph"): fl |= ts.FLAG_NO_GRAPH
    e = ts.Engine(n, l, k, flags=fl)
    e.synth_genotypes(theta, beta, seed=3); e.set_gamma(gamma)
    e.run_schedule(locs); e.synchronize()
    out[mode] = (e.get_lambda(), e.get_gamma(), e.total_passes()); e.close()
def rel(a, b): return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-300)))
for m in ("split", "split-nograph"):
    print(m, "lambda rel", rel(out[m][0], out["single"][0]), "gamma rel", rel(out[m][1], out["single"][1]), "passes", out[m][2], out["single"][2])
