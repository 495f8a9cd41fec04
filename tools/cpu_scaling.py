"""How the CPU restatement (oracle/, OpenMP in the reference's work partition) scales with threads on
this host: python tools/cpu_scaling.py [N] -- prints updates/s per thread count."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_py as op
from helpers import pack_bed, psd_genotypes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
k, l = 8, 4
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
y, _, _ = psd_genotypes(n, l, k, 1)
payload = pack_bed(y)
g = np.random.default_rng(2).gamma(100.0, 0.01, size=(n, k))
for t in [1, 4, 8, 16, 32, 64, 128, 256]:
    if t > (os.cpu_count() or 1):
        break
    orc = op.Oracle(n, l, k, nthreads=t)
    orc.load_bed_payload(payload)
    orc.set_gamma(g)
    orc.snp_update(0)
    t0 = time.perf_counter()
    reps = 1 if t < 8 else 3
    for i in range(reps):
        orc.snp_update((i + 1) % l)
    dt = (time.perf_counter() - t0) / reps
    print(f"threads {t:4d}: {dt*1e3:9.1f} ms/update  {1/dt:8.3f} updates/s", flush=True)
    orc.close()
