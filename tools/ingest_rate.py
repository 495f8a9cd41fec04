"""Host -> HBM ingest rate of tsamd_upload_bed (PCIe-inclusive), and an end-to-end CLI run
on a synthetic .bed (config 2 of BASELINE.json: N=10K, L=100K, K=6)."""
import os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import terastructure_amd as ts

n, l, k = 1_000_000, 16384, 8
payload = np.random.default_rng(0).integers(0, 256, size=(l, (n + 3) // 4), dtype=np.uint8)
with ts.Engine(n, l, k) as e:
    e.upload_bed(payload[:64])
    t0 = time.perf_counter(); e.upload_bed(payload); dt = time.perf_counter() - t0
    print(f"upload_bed: {payload.nbytes/1e9:.2f} GB in {dt:.3f} s = {payload.nbytes/dt/1e9:.1f} GB/s (pageable host buffer, staged through 32 MB pinned chunks)")
del payload
# config 2 through the CLI
n, l, k = 10_000, 100_000, 6
d = "/tmp/ts_cfg2"; os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(1)
theta = rng.dirichlet(np.full(k, 0.2), size=n); beta = rng.uniform(0.05, 0.95, size=(l, k))
CODE = np.array([0, 2, 3], dtype=np.uint8)
with open(f"{d}/syn.bed", "wb") as f:
    f.write(bytes([0x6C, 0x1B, 0x01]))
    for l0 in range(0, l, 5000):
        p = beta[l0:l0+5000] @ theta.T
        y = (rng.random(p.shape) < p).astype(np.uint8) + (rng.random(p.shape) < p).astype(np.uint8)
        c = CODE[y].reshape(p.shape[0], n // 4, 4)
        f.write((c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).astype(np.uint8).tobytes())
open(f"{d}/syn.bim", "w").write("".join(f"1\ts{i}\t0\t{i}\tA\tB\n" for i in range(l)))
open(f"{d}/syn.fam", "w").write("".join(f"{i} {i} 0 0 0 -9\n" for i in range(n)))
t0 = time.perf_counter()
r = subprocess.run([os.path.join(R, "host", "terastructure"), "-file", "syn.bed", "-n", str(n), "-l", str(l), "-k", str(k),
                    "-rfreq", "10000", "-seed", "7", "-label", "cfg2", "-force", "-max-iter", "60000"], cwd=d, capture_output=True, text=True)
dt = time.perf_counter() - t0
print("cli rc", r.returncode, f"wall {dt:.1f} s for 60000 iterations + 6 validation passes (N=10K, L=100K, K=6)")
print(open(f"{d}/n10000-k6-l100000-cfg2-seed7/validation.txt").read())
th = np.loadtxt(f"{d}/n10000-k6-l100000-cfg2-seed7/theta.txt")
import itertools
best = min(np.sqrt(np.mean((th[:, list(pm)] - theta) ** 2)) for pm in itertools.permutations(range(k)))
print("best-permutation RMSE(theta, truth) after 60K iterations:", best)
