"""Where a short schedule's fixed cost goes (the driver's bench line times 20 updates in one call): per call of
run_schedule(n) + synchronize -- host time to enqueue, host time waiting, and the launch's own duration (HIP events) --
for n = 1, 5, 20, 100, 400.   usage: python3 tools/short_run_overhead.py [N] [K]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import terastructure_amd as ts  # noqa: E402
from helpers import init_gamma  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
l = 64
torch.cuda.init()
eng = ts.Engine(n, l, k)
rng = np.random.default_rng(1)
eng.upload_bed(rng.integers(0, 256, size=(l, (n + 3) // 4), dtype=np.uint8) & 0xBB | 0)  # (no 01 = missing codes needed here)
eng.set_gamma(init_gamma(n, k, 3))
eng.prepare()
eng.run_schedule(rng.integers(0, l, size=300).astype(np.uint32))
eng.synchronize()
for cnt in (1, 5, 20, 100, 400):
    rows = []
    for rep in range(12):
        locs = rng.integers(0, l, size=cnt).astype(np.uint32)
        eng.synchronize()
        torch.cuda.synchronize()
        eng.profile_enable(True)
        t0 = time.perf_counter()
        eng.run_schedule(locs)
        t1 = time.perf_counter()
        eng.synchronize()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        pr = eng.profile_read()
        eng.profile_enable(False)
        rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, pr.get("pass_ms", 0.0) * 1e3 if isinstance(pr, dict) else 0.0))
    r = np.median(np.array(rows[2:]), axis=0)
    print(f"n={cnt:4d}: enqueue {r[0]:7.1f} us  wait {r[1]:8.1f} us  torch sync {r[2]:6.1f} us  total {r[0]+r[1]+r[2]:8.1f}  "
          f"per update {(r[0]+r[1]+r[2])/cnt:7.2f} | launch by events {r[3]:8.1f} us ({r[3]/cnt:6.2f} per update)  {pr}")
eng.close()
