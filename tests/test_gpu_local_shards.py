"""Several shards driven by ONE process (tsamd_p2p_connect_local): the way the drop-in host
uses all GPUs of a node from its main thread.  On the one-GPU test box all shards share
device 0, which exercises everything except the xGMI hop."""
import numpy as np
import pytest

import oracle_py as op
from helpers import init_gamma, pack_bed, psd_genotypes, rel_err
from test_gpu_parity import ts  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,n,k", [(2, 3001, 5), (3, 20000, 8), (4, 5000, 20)])
def test_local_shards_match_oracle(ts, world, n, k):
    l, seed = 24, 300 + world
    y, _, _ = psd_genotypes(n, l, k, seed, 0.02)
    payload = pack_bed(y)
    gamma = init_gamma(n, k, seed + 1)
    orc = op.Oracle(n, l, k)
    orc.load_bed_payload(payload)
    orc.set_gamma(gamma)
    engs = [ts.Engine(n, l, k, device=0, rank=r, world=world) for r in range(world)]
    try:
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
        # held-out log likelihood: the shards' sums add up to the oracle's
        for loc in held:
            parts = [e.heldout_loglik(loc) for e in engs]
            s, cnt = sum(p[0] for p in parts), sum(p[1] for p in parts)
            so, co = orc.heldout_loglik(loc)
            assert cnt == co and abs(s - so) <= 1e-9 * abs(so)
    finally:
        for e in engs:
            e.close()


def test_local_connect_errors(ts):
    a = ts.Engine(1000, 8, 3, rank=0, world=2)
    b = ts.Engine(1000, 8, 3, rank=0, world=2)   # same rank twice
    try:
        with pytest.raises(ts.TsamdError):
            ts.Engine.p2p_connect_local([a, b])
        with pytest.raises(ts.TsamdError):
            ts.Engine.p2p_connect_local([a])       # world is 2
    finally:
        a.close()
        b.close()


def test_local_shards_deep_queue(ts):
    """Far more kernels than a device queue holds (pass cap 100 as in -compute-beta, 1500
    locations): run_schedule_all interleaves the shards in bounded batches, so the one driving
    thread never blocks with a peer's work unsubmitted."""
    world, n, l, k = 2, 2000, 1500, 4
    y, _, _ = psd_genotypes(n, l, k, 77, 0.01)
    payload = pack_bed(y)
    gamma = init_gamma(n, k, 78)
    engs = [ts.Engine(n, l, k, device=0, rank=r, world=world, max_inner=100) for r in range(world)]
    ref = ts.Engine(n, l, k, max_inner=100)
    try:
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
    finally:
        for e in engs + [ref]:
            e.close()
