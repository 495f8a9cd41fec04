#!/usr/bin/env python3
"""Write a synthetic PLINK .bed / .bim / .fam of the PSD model (theta_n ~ Dirichlet(0.2), beta_lk ~ U(0.05, 0.95),
y ~ Binomial(2, theta_n . beta_l)) -- generated on the GPU (tsamd_synth_genotypes, the generator bench.py uses) chunk by chunk
and read back column by column, because numpy would take tens of minutes for 5e10 genotypes.  For end-to-end runs of the host
CLI at BASELINE sizes:  tools/make_synth_bed.py out_prefix N L K [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    prefix, n, l, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 7
    import terastructure_amd as ts

    rng = np.random.default_rng(seed)
    theta = rng.dirichlet(np.full(k, 0.2), size=n)
    chunk = max(1, min(l, (1 << 30) // max(1, (n + 3) // 4)))   # about 1 GB of columns per engine
    t0 = time.time()
    with open(prefix + ".bed", "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))
        for l0 in range(0, l, chunk):
            nl = min(chunk, l - l0)
            beta = rng.uniform(0.05, 0.95, size=(nl, k))
            with ts.Engine(n, nl, k) as eng:
                eng.synth_genotypes(theta, beta, 0, seed + 1 + l0, 0.0)
                eng.synchronize()
                buf = bytearray()
                for loc in range(nl):
                    buf += eng.download_bed(loc).tobytes()
                    if len(buf) >= (64 << 20):
                        f.write(buf)
                        buf = bytearray()
                f.write(buf)
            print(f"[make_synth_bed] {l0 + nl} of {l} columns, {time.time() - t0:.1f} s", flush=True)
    for ext, count in ((".bim", l), (".fam", n)):   # only line-counted by the reference (src/snp.cc:104-139)
        with open(prefix + ext, "w") as f:
            f.write("x\n" * count)
    print(f"[make_synth_bed] wrote {prefix}.bed: {os.path.getsize(prefix + '.bed') / 1e9:.2f} GB in {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main()
