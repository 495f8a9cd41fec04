"""Randomised parity stress (tests/stress_cases.py; a fixed-seed 8-case slice of it runs in the suite,
tests/test_gpu_geometry.py::test_random_geometries_slice).  python tools/stress_parity.py [cases] [seed]"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import terastructure_amd as ts
from stress_cases import run_case

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    ok, desc = run_case(ts, rng)
    if not ok:
        bad += 1
        print(f"MISMATCH case {c}: {desc}", flush=True)
print(f"{cases} cases, {bad} mismatches")
