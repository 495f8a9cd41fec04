"""Several shards on ONE GPU driven by one process (tsamd_p2p_connect_local): what the
peer-to-peer protocol does functionally at scale.  NOT a performance proxy: kernels of two queues
that spin on each other share one GPU badly (measured: 2 shards of N=1M on one MI355X run 7x
slower than one shard, ~100 us per pass; with 8 the 3 s peer timeout fires).
python tools/local_shards_bench.py [world] [N]   (compare with world = 1)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import terastructure_amd as ts

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
l, k, steps = 2000, 8, 1500
rng = np.random.default_rng(1)
theta = rng.dirichlet(np.full(k, 0.2), size=n)
beta = rng.uniform(0.05, 0.95, size=(l, k))
gamma = rng.gamma(100.0, 0.01, size=(n, k))
locs = rng.integers(0, l, size=steps + 100).astype(np.uint32)
engs = [ts.Engine(n, l, k, rank=r, world=world) for r in range(world)]
for e in engs:
    b, c = e.shard_begin, e.shard_count
    e.synth_genotypes(theta[b:b + c], beta, seed=3)
    e.set_gamma(gamma[b:b + c])
if world > 1:
    ts.Engine.p2p_connect_local(engs)
def run(part):
    if world > 1:
        ts.Engine.run_schedule_all(engs, part)
    else:
        engs[0].run_schedule(part)
    for e in engs:
        e.synchronize()
run(locs[:100])
t0 = time.perf_counter()
run(locs[100:])
dt = time.perf_counter() - t0
print(f"world {world} N {n}: {steps/dt:9.1f} updates/s  ({dt/steps*1e6:.1f} us/update)")
for e in engs:
    e.close()
