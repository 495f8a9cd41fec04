"""Minimax polynomial P(f), f = 1/z^2, for the tail of psi(z) - log z = -1/(2z) + f P(f) on z >= 10 (f <= 0.01): the absolute
error of f P(f) is levelled (it adds to an exponent), Remez exchange in 60-digit arithmetic (mpmath).  For
exp_digamma_split in terastructure_amd/csrc/tsamd_device.h, which used the asymptotic series -1/12 + f/120 - f^2/252 + ...
(seven terms).   usage: python3 tools/fit/psi_tail_minimax.py [degree]"""
import sys

import mpmath as mp

mp.mp.dps = 60
DEG = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = mp.mpf("0.01")


def h(f):
    z = 1 / mp.sqrt(f)
    return (mp.digamma(z) - mp.log(z) + 1 / (2 * z)) / f


def fit(nodes):
    n = DEG + 2
    m = mp.matrix(n, n)
    rhs = mp.matrix(n, 1)
    for i, x in enumerate(nodes):
        for j in range(DEG + 1):
            m[i, j] = x ** j
        m[i, DEG + 1] = -((-1) ** i) / x
        rhs[i] = h(x)
    sol = mp.lu_solve(m, rhs)
    return [sol[j] for j in range(DEG + 1)], sol[DEG + 1]


def err(c, x):
    return x * (mp.polyval(c[::-1], x) - h(x))


NG = 3000
grid = [B * mp.mpf(i) / NG for i in range(1, NG + 1)]
hv = None
nodes = [B * (1 + mp.cos(mp.pi * (DEG + 1 - i) / (DEG + 1.5))) / 2 for i in range(DEG + 2)]
nodes = [max(x, B / 1000) for x in nodes]
for _ in range(15):
    c, e = fit(nodes)
    vals = [err(c, x) for x in grid]
    ext = []
    for i in range(len(grid)):
        lo = abs(vals[i - 1]) if i > 0 else mp.mpf(0)
        hi = abs(vals[i + 1]) if i + 1 < len(grid) else mp.mpf(0)
        if abs(vals[i]) >= lo and abs(vals[i]) >= hi:
            ext.append(i)
    keep = []
    for i in ext:
        if keep and mp.sign(vals[keep[-1]]) == mp.sign(vals[i]):
            if abs(vals[i]) > abs(vals[keep[-1]]):
                keep[-1] = i
        else:
            keep.append(i)
    if len(keep) < DEG + 2:
        break
    while len(keep) > DEG + 2:
        keep.pop(0 if abs(vals[keep[0]]) < abs(vals[keep[-1]]) else -1)
    nodes = [grid[i] for i in keep]
c, e = fit(nodes)
print(f"degree {DEG} in f on (0, {B}]: levelled absolute error of f P(f): {mp.nstr(abs(e), 5)}")
cd = [float(x) for x in c]
worst = max(abs(err([mp.mpf(v) for v in cd], x)) for x in grid)
print(f"coefficients rounded to double: max absolute error {mp.nstr(worst, 5)} (exact evaluation)")
for j, v in enumerate(cd):
    print(f"  t{j:<2d} = {v.hex():>26s}   // {v!r}")
