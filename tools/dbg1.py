import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import terastructure_amd as ts
import oracle_py as op
from helpers import psd_genotypes, pack_bed, init_gamma
for n in (64, 512, 2500):
    k, l = 2, 4
    y, _, _ = psd_genotypes(n, l, k, 3, 0.0)
    pl = pack_bed(y); g = init_gamma(n, k, 4)
    orc = op.Oracle(n, l, k, online_iterations=1); orc.load_bed_payload(pl); orc.set_gamma(g)
    with ts.Engine(n, l, k, max_inner=1) as e:
        e.upload_bed(pl); e.set_gamma(g)
        e.snp_update(0); orc.snp_update(0)
        print(n, "dev", e.get_lambda(0,1)[0].ravel(), "orc", orc.lambda_()[0].ravel())
