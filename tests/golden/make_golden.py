"""Regenerates tests/golden/config1_oracle.npz from the CPU oracle (oracle/ts_oracle.c) on the
reference's data/test.bed with data/run.sh's flags (-n 200 -l 10000 -k 3 -seed 1234 -rfreq 1000).

    python tests/golden/make_golden.py

The file pins the oracle itself (tests/test_oracle_golden.py::test_oracle_matches_committed_golden)
and gives the GPU parity test fixed expected values (tests/test_gpu_parity.py).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_py as op  # noqa: E402

n, l, k = 200, 10000, 3
bed = os.path.join(HERE, "ref_data", "test.bed")
L = op.lib()

# (a) first report period, the exact stream of the reference run
orc = op.Oracle(n, l, k)
orc.read_bed_file(bed)
r = op.gsl_mt19937(1234)
L.orc_set_validation_sample(orc.s, C.byref(r))
L.orc_init_gamma(orc.s, C.byref(r))
gamma0 = orc.gamma()
held_locs = orc.heldout_locs().astype(np.uint32)
held_indivs = np.stack([orc.heldout_indivs(int(x)) for x in held_locs]).astype(np.uint32)
locs = np.array([L.orc_rng_uniform_int(C.byref(r), l) for _ in range(1000)], dtype=np.uint32)
its = np.array([orc.snp_update(int(x)) for x in locs], dtype=np.uint32)
val_ll = []
for x in held_locs:
    orc.snp_update(int(x), 1)
    val_ll.append(orc.heldout_loglik(int(x))[0])
gamma1050 = orc.gamma()
theta1050 = orc.theta()
lam_sample = orc.lambda_()[locs[:8]]

# (b) the whole run to its own stop rule
full = op.Oracle(n, l, k)
full.read_bed_file(bed)
res = full.run(seed=1234, reportfreq=1000)

np.savez_compressed(os.path.join(HERE, "config1_oracle.npz"),
                    gamma0=gamma0, held_locs=held_locs, held_indivs=held_indivs, locs=locs, inner_iters=its,
                    val_ll=np.array(val_ll), gamma1050=gamma1050, theta1050=theta1050, lam_sample=lam_sample,
                    final_theta=full.theta(), final_gamma=full.gamma(), final_iter=res["final_iter"],
                    val_iters=np.array([x[0] for x in res["lines"]]), val_mean_ll=np.array([x[1] for x in res["lines"]]))
print("wrote config1_oracle.npz")
