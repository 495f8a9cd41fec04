import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import terastructure_amd as ts
import oracle_py as op
from helpers import psd_genotypes, pack_bed, init_gamma, rel_err
n, l, k = 3000, 32, 6
y, _, _ = psd_genotypes(n, l, k, 55, 0.02)
pl = pack_bed(y); g = init_gamma(n, k, 56)
locs = np.random.default_rng(4).integers(0, l, size=20).astype(np.uint32)
orc = op.Oracle(n, l, k); orc.load_bed_payload(pl); orc.set_gamma(g)
for loc in locs: orc.snp_update(int(loc))
for name, flags in (("fused", 0), ("split", ts.FLAG_SPLIT_EPILOGUE), ("split-nograph", ts.FLAG_SPLIT_EPILOGUE | ts.FLAG_NO_GRAPH), ("fused-nograph", ts.FLAG_NO_GRAPH)):
    with ts.Engine(n, l, k, flags=flags) as e:
        e.upload_bed(pl); e.set_gamma(g)
        e.run_schedule(locs); e.synchronize()
        print(name, "lambda err", rel_err(e.get_lambda(), orc.lambda_()), "gamma err", rel_err(e.get_gamma(), orc.gamma()), "passes", e.total_passes())
