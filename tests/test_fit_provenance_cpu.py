"""The two fitted polynomials in csrc/tsamd_device.h are what tools/fit/*.py produce (Remez exchange in 60-digit
arithmetic), and they are as accurate as the comments next to them say."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEVICE_H = os.path.join(ROOT, "terastructure_amd", "csrc", "tsamd_device.h")

mpmath = pytest.importorskip("mpmath")


def run_fit(script, degree):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fit", script), str(degree)], capture_output=True, text=True,
                         timeout=600, check=True).stdout
    coeffs = re.findall(r"=\s+(-?0x[0-9a-f.]+p[+-]?\d+)", out)
    err = float(re.search(r"rounded to double: max \w+ error ([0-9.e+-]+)", out).group(1))
    return coeffs, err


def literals_in_header():
    return set(re.findall(r"-?0x1\.[0-9a-f]+p[+-]?\d+", open(DEVICE_H).read()))


def test_exp_polynomial_is_the_degree_11_fit():
    coeffs, err = run_fit("exp_minimax.py", 11)
    assert len(coeffs) == 12 and err < 2e-17
    have = literals_in_header()
    for c in coeffs[2:]:          # (c0 = c1 = 1.0 are written as such)
        assert c in have, f"{c} of the exp fit is not in tsamd_device.h"
    assert float.fromhex(coeffs[0]) == 1.0 and float.fromhex(coeffs[1]) == 1.0


def test_digamma_tail_polynomial_is_the_degree_4_fit():
    coeffs, err = run_fit("psi_tail_minimax.py", 4)
    assert len(coeffs) == 5 and err < 1.1e-17
    have = literals_in_header()
    for c in coeffs:
        assert c in have or c.lstrip("-") in {h.lstrip("-") for h in have}, f"{c} of the tail fit is not in tsamd_device.h"


def test_digamma_recurrence_coefficients():
    """Q(q) = prod_{i<5} (q + i (9 - i)) and its derivative, expanded: the integer coefficients exp_digamma_split uses"""
    import numpy as np

    poly = np.poly1d([1.0])
    for i in range(5):
        poly = poly * np.poly1d([1.0, float(i * (9 - i))])
    assert [int(round(c)) for c in poly.coeffs] == [1, 60, 1308, 12176, 40320, 0]
    assert [int(round(c)) for c in poly.deriv().coeffs] == [5, 240, 3924, 24352, 40320]
    src = open(DEVICE_H).read()
    for c in ("60.0", "1308.0", "12176.0", "40320.0", "240.0", "3924.0", "24352.0"):
        assert c in src
