"""Shared builders for parity tests: seeded PSD genotypes as PLINK payloads."""
import os

import numpy as np

# PLINK 2-bit code of genotype y (count of A2): 0 -> 00, 1 -> 10, 2 -> 11, missing -> 01
# (decode rule src/snp.cc:203-216)
CODE_OF_Y = np.array([0b00, 0b10, 0b11, 0b01], dtype=np.uint8)
Y_OF_CODE = np.array([0, 3, 1, 2], dtype=np.uint8)


def psd_genotypes(n, l, k, seed, missing_rate=0.0):
    """y[l][n] in {0,1,2,3}; Pritchard-Stephens-Donnelly model."""
    rng = np.random.default_rng(seed)
    theta = rng.dirichlet(np.full(k, 0.2), size=n)
    beta = rng.uniform(0.05, 0.95, size=(l, k))
    p = beta @ theta.T  # [l][n]
    y = (rng.random((l, n)) < p).astype(np.uint8) + (rng.random((l, n)) < p).astype(np.uint8)
    if missing_rate > 0:
        y[rng.random((l, n)) < missing_rate] = 3
    return y, theta, beta


def pack_bed(y):
    """y[l][n] -> PLINK SNP-major payload [l][ceil(n/4)] (zero padding bits, like PLINK)."""
    l, n = y.shape
    nb = (n + 3) // 4
    codes = np.zeros((l, nb * 4), dtype=np.uint8)
    codes[:, :n] = CODE_OF_Y[y]
    c = codes.reshape(l, nb, 4)
    return (c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).astype(np.uint8)


def unpack_bed(payload, n):
    l, nb = payload.shape
    c = np.stack([(payload >> (2 * j)) & 3 for j in range(4)], axis=2).reshape(l, nb * 4)[:, :n]
    return Y_OF_CODE[c]


def init_gamma(n, k, seed):
    """Gamma(100, 0.01) rows like init_gamma (src/snpsamplinge.cc:226-237), any RNG."""
    return np.random.default_rng(seed).gamma(100.0, 0.01, size=(n, k))


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-300)))


def usable_cores():
    """Host cores this process may really use: the smaller of the affinity mask and the
    cgroup CPU quota (a container can see 256 CPUs and be entitled to 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:  # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def slow_params(values, fast):
    """pytest parametrize values: those not in `fast` carry the `slow` marker (tests/conftest.py skips them unless TS_RUN_SLOW=1)"""
    import pytest

    return [v if v in fast else pytest.param(v, marks=pytest.mark.slow) for v in values]
