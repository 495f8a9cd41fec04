"""Minimax polynomial of degree 11 for exp(r) on |r| <= A (relative error), by the Remez exchange in 60-digit arithmetic
(mpmath), for exp_nonpos in terastructure_amd/csrc/tsamd_device.h.  Prints the coefficients as C hex-float literals
with the error of the ROUNDED polynomial evaluated in exact arithmetic.   usage: python3 tools/fit/exp_minimax.py [degree]"""
import sys

import mpmath as mp

mp.mp.dps = 60
DEG = int(sys.argv[1]) if len(sys.argv) > 1 else 11
A = mp.mpf("0.3475")  # ln(2)/2 = 0.34657..., with room for the reduction's rounding


def f(x):
    return mp.e ** x


def fit(nodes):
    """coefficients c_0..c_DEG and E with sum c_j x^j - (-1)^i E f(x_i) = f(x_i): the relative error alternates"""
    n = DEG + 2
    m = mp.matrix(n, n)
    rhs = mp.matrix(n, 1)
    for i, x in enumerate(nodes):
        for j in range(DEG + 1):
            m[i, j] = x ** j
        m[i, DEG + 1] = -((-1) ** i) * f(x)
        rhs[i] = f(x)
    sol = mp.lu_solve(m, rhs)
    return [sol[j] for j in range(DEG + 1)], sol[DEG + 1]


def rel_err(c, x):
    return mp.polyval(c[::-1], x) / f(x) - 1


nodes = [A * mp.cos(mp.pi * (DEG + 1 - i) / (DEG + 1)) for i in range(DEG + 2)]  # Chebyshev extrema, ascending
for _ in range(12):
    c, e = fit(nodes)
    # new nodes: the extrema of the error between consecutive zeros (dense search + refinement)
    grid = [-A + 2 * A * mp.mpf(i) / 4000 for i in range(4001)]
    vals = [rel_err(c, x) for x in grid]
    ext = []
    for i in range(1, 4000):
        if (abs(vals[i]) >= abs(vals[i - 1]) and abs(vals[i]) >= abs(vals[i + 1])):
            ext.append(grid[i])
    ext = [grid[0]] + ext + [grid[-1]]
    # keep DEG + 2 alternating extrema with the largest magnitudes
    keep = []
    for x in ext:
        v = rel_err(c, x)
        if keep and mp.sign(rel_err(c, keep[-1])) == mp.sign(v):
            if abs(v) > abs(rel_err(c, keep[-1])):
                keep[-1] = x
        else:
            keep.append(x)
    if len(keep) < DEG + 2:
        break
    while len(keep) > DEG + 2:
        keep.pop(0 if abs(rel_err(c, keep[0])) < abs(rel_err(c, keep[-1])) else -1)
    nodes = keep
c, e = fit(nodes)
print(f"degree {DEG} on |r| <= {A}: levelled relative error {mp.nstr(abs(e), 5)}")
cd = [float(x) for x in c]
worst = max(abs(mp.polyval([mp.mpf(v) for v in cd][::-1], x) / f(x) - 1) for x in [-A + 2 * A * mp.mpf(i) / 8000 for i in range(8001)])
print(f"coefficients rounded to double: max relative error {mp.nstr(worst, 5)} (exact evaluation)")
for j, v in enumerate(cd):
    print(f"  c{j:<2d} = {v.hex():>24s}   // {v!r}")
