"""Worker of test_gpu_local_shards.py: runs in a fresh process, because several contexts on ONE
device need one hardware queue each (HIP maps streams onto GPU_MAX_HW_QUEUES queues round-robin;
two shards whose kernels spin on each other in the same queue never finish).  On a node with
one shard per GPU the question does not arise."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import oracle_py as op  # noqa: E402
from helpers import init_gamma, pack_bed, psd_genotypes, rel_err  # noqa: E402


def match_oracle(ts, world, n, k):
    l, seed = 24, 300 + world
    y, _, _ = psd_genotypes(n, l, k, seed, 0.02)
    payload = pack_bed(y)
    gamma = init_gamma(n, k, seed + 1)
    orc = op.Oracle(n, l, k)
    orc.load_bed_payload(payload)
    orc.set_gamma(gamma)
    engs = [ts.Engine(n, l, k, device=0, rank=r, world=world) for r in range(world)]
    rng = np.random.default_rng(seed + 2)
    held = {}
    for loc in rng.choice(l, size=3, replace=False):
        cand = np.nonzero(y[loc] != 3)[0]
        held[int(loc)] = np.sort(rng.choice(cand, size=max(1, n // 40), replace=False)).astype(np.uint32)
        orc.set_heldout(int(loc), held[int(loc)])
    for e in engs:
        e.upload_bed(payload)
        e.set_gamma(gamma[e.shard_begin:e.shard_begin + e.shard_count])
        for loc, ids in held.items():
            e.set_heldout(loc, ids)
    ts.Engine.p2p_connect_local(engs)
    locs = np.random.default_rng(seed + 3).integers(0, l, size=50).astype(np.uint32)
    for part, hol in ((locs[:7], 0), (locs[7:8], 1), (locs[8:], 0)):   # eager, held-out mode, graph replay
        ts.Engine.run_schedule_all(engs, part, hol_mode=hol)
        for e in engs:
            e.synchronize()
    its = [orc.snp_update(int(loc), 1 if i == 7 else 0) for i, loc in enumerate(locs)]
    g = np.concatenate([e.get_gamma() for e in engs])
    c = np.concatenate([e.get_counts() for e in engs])
    assert rel_err(g, orc.gamma()) < 1e-9
    assert np.array_equal(c, orc.c_indiv())
    for e in engs:
        assert rel_err(e.get_lambda(), orc.lambda_()) < 1e-9
        assert e.total_passes() == sum(its)
        assert np.array_equal(e.get_lambda(), engs[0].get_lambda())  # replicated state: same bits
    for loc in held:   # held-out log likelihood: the shards' sums add up to the oracle's
        parts = [e.heldout_loglik(loc) for e in engs]
        s, cnt = sum(p[0] for p in parts), sum(p[1] for p in parts)
        so, co = orc.heldout_loglik(loc)
        assert cnt == co and abs(s - so) <= 1e-9 * abs(so)
    for e in engs:
        e.close()


def deep_queue(ts):
    world, n, l, k = 2, 2000, 1500, 4
    y, _, _ = psd_genotypes(n, l, k, 77, 0.01)
    payload = pack_bed(y)
    gamma = init_gamma(n, k, 78)
    engs = [ts.Engine(n, l, k, device=0, rank=r, world=world, max_inner=100) for r in range(world)]
    ref = ts.Engine(n, l, k, max_inner=100)
    for e in engs + [ref]:
        e.upload_bed(payload)
        e.set_gamma(gamma[e.shard_begin:e.shard_begin + e.shard_count])
    ts.Engine.p2p_connect_local(engs)
    locs = np.arange(l, dtype=np.uint32)
    ts.Engine.run_schedule_all(engs, locs)
    for e in engs:
        e.synchronize()
    ref.run_schedule(locs)
    ref.synchronize()
    assert engs[0].total_passes() == ref.total_passes()
    assert rel_err(engs[0].get_lambda(), ref.get_lambda()) < 1e-10
    assert rel_err(np.concatenate([e.get_gamma() for e in engs]), ref.get_gamma()) < 1e-10
    for e in engs + [ref]:
        e.close()


def connect_errors(ts):
    a = ts.Engine(1000, 8, 3, rank=0, world=2)
    b = ts.Engine(1000, 8, 3, rank=0, world=2)   # same rank twice
    for group in ([a, b], [a]):                   # ... and a world of 2 given one context
        try:
            ts.Engine.p2p_connect_local(group)
        except ts.TsamdError:
            pass
        else:
            raise AssertionError("p2p_connect_local accepted an inconsistent group")
    a.close()
    b.close()


if __name__ == "__main__":
    import terastructure_amd as ts

    what = sys.argv[1]
    if what == "match":
        match_oracle(ts, *(int(x) for x in sys.argv[2:5]))
    elif what == "deep":
        deep_queue(ts)
    else:
        connect_errors(ts)
    print("worker ok")
