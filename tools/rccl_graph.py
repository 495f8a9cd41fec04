"""Single GPU, world = 1 communicator: the RCCL exchange path launched eagerly vs captured in the
hipGraph (TSAMD_RCCL_GRAPH=1).  python tools/rccl_graph.py [N]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import terastructure_amd as ts

n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
l, k = 2000, 8
rng = np.random.default_rng(1)
theta = rng.dirichlet(np.full(k, 0.2), size=n)
beta = rng.uniform(0.05, 0.95, size=(l, k))
gamma = rng.gamma(100.0, 0.01, size=(n, k))
locs = rng.integers(0, l, size=2100).astype(np.uint32)
res = {}
for mode in ("single", "rccl-eager", "rccl-graph"):
    os.environ["TSAMD_RCCL_GRAPH"] = "1" if mode == "rccl-graph" else "0"
    e = ts.Engine(n, l, k, flags=0 if mode == "single" else ts.FLAG_SPLIT_EPILOGUE)
    e.synth_genotypes(theta, beta, seed=3)
    e.set_gamma(gamma)
    if mode != "single":
        e.comm_init(e.comm_unique_id())
    e.run_schedule(locs[:100]); e.synchronize()
    t0 = time.perf_counter()
    e.run_schedule(locs[100:]); e.synchronize()
    dt = time.perf_counter() - t0
    res[mode] = (e.get_lambda(), e.get_gamma())
    print(f"{mode:11s}: {2000/dt:9.1f} updates/s  ({dt/2000*1e6:.1f} us/update)", flush=True)
    e.close()
for m in ("rccl-eager", "rccl-graph"):
    print(m, "max rel diff to single:", max(float(np.max(np.abs(a - b) / (np.abs(b) + 1e-300))) for a, b in zip(res[m], res["single"])))
