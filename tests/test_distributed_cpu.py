"""world_size-2 gloo tests (CPU) of the sharded protocol: individuals split by
tsamd_shard_range, one all-reduce(sum) of the 2K lambda statistics per pass,
the K x 2 epilogue and the convergence decision replicated on every rank, the
gamma step shard-local.  The per-shard arithmetic is the oracle's; what is under
test is the partition, the exchange and terastructure_amd.dist."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_py as op
from helpers import init_gamma, pack_bed, psd_genotypes, rel_err


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class ShardedOracle:
    """Rank-local stand-in for an Engine shard: oracle partial sums over the rank's
    individuals + gloo all-reduce where libtsamd uses RCCL."""

    def __init__(self, n, l, k, payload, gamma, rank, world, shard_range):
        self.orc = op.Oracle(n, l, k)
        self.orc.load_bed_payload(payload)
        self.orc.set_gamma(gamma)
        self.b, self.c = shard_range(n, rank, world)
        self.k = k
        self.pending = None

    def snp_update(self, loc, hol=0):
        o = self.orc
        if self.pending is not None and not self.pending[1]:
            o.gamma_step(self.pending[0])          # every rank holds all rows here; only
        self.pending = (loc, hol)                  # its own shard's rows are compared
        it = 0
        while True:
            part = o.pass_partial(loc, self.b, self.b + self.c)
            t = torch.from_numpy(part.reshape(-1).copy())
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            v = o.epilogue(loc, t.numpy().reshape(self.k, 2))
            it += 1
            if v < 1e-3 or it >= 10:
                return it


class GroupedShardedOracle(ShardedOracle):
    """The in-launch exchange of ts_schedule on a shard, in the oracle's arithmetic: the rank's
    individuals are cut into workgroup chunks, workgroup w belongs to group w % 8, a group's leader adds its members'
    partial rows in member order, the group sums of ALL ranks are gathered (the kernel: stored into every rank's
    Xchg::res_sums) and every rank adds them in (rank, group) order.  No reduction tree decides the order, so every
    rank gets the same bits."""

    GROUPS = 8

    def __init__(self, *a, chunk=64, **kw):
        super().__init__(*a, **kw)
        self.chunk = chunk

    def snp_update(self, loc, hol=0):
        o = self.orc
        if self.pending is not None and not self.pending[1]:
            o.gamma_step(self.pending[0])
        self.pending = (loc, hol)
        world = dist.get_world_size()
        it = 0
        while True:
            sums = np.zeros((self.GROUPS, self.k, 2))
            for w, b in enumerate(range(self.b, self.b + self.c, self.chunk)):   # member order within a group
                sums[w % self.GROUPS] += o.pass_partial(loc, b, min(b + self.chunk, self.b + self.c))
            gathered = [torch.zeros(sums.size, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(gathered, torch.from_numpy(sums.reshape(-1).copy()))
            total = np.zeros((self.k, 2))
            for r in range(world):                                               # (rank, group) order
                rows = gathered[r].numpy().reshape(self.GROUPS, self.k, 2)
                for g in range(self.GROUPS):
                    total += rows[g]
            v = o.epilogue(loc, total)
            it += 1
            if v < 1e-3 or it >= 10:
                return it


def _worker(rank, world, port, n, l, k, seed, out_dir, grouped=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import terastructure_amd as ts
    from terastructure_amd import dist as tdist

    d, r, w = tdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    y, _, _ = psd_genotypes(n, l, k, seed, 0.03)
    payload = pack_bed(y)
    gamma = init_gamma(n, k, seed + 1)
    sh = (GroupedShardedOracle if grouped else ShardedOracle)(n, l, k, payload, gamma, rank, world, ts.shard_range)
    locs = np.random.default_rng(seed + 2).integers(0, l, size=12)
    its = [sh.snp_update(int(loc)) for loc in locs]
    # phi of the other shard's individuals was never computed on this rank, so only the
    # local rows of gamma are meaningful: gather them like the host does for theta.txt
    local = sh.orc.gamma()[sh.b:sh.b + sh.c]
    full = tdist.gather_rows(local, n, d, ts.shard_range)
    s = tdist.sum_over_ranks([float(sh.c), 1.0], d)
    assert s == [float(n), float(world)]
    assert tdist.max_over_ranks(rank, d) == world - 1
    uid = [b"x" * 128 if rank == 0 else None]
    d.broadcast_object_list(uid, src=0)         # the side channel bootstrap_comm uses
    assert uid[0] == b"x" * 128
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), its=np.array(its), lam=sh.orc.lambda_(), gamma=full)
    d.barrier()
    d.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_protocol_matches_single_rank(tmp_path, world):
    n, l, k, seed = 1003, 24, 5, 41
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, l, k, seed, str(tmp_path)), nprocs=world, join=True)
    y, _, _ = psd_genotypes(n, l, k, seed, 0.03)
    ref = op.Oracle(n, l, k)
    ref.load_bed_payload(pack_bed(y))
    ref.set_gamma(init_gamma(n, k, seed + 1))
    locs = np.random.default_rng(seed + 2).integers(0, l, size=12)
    its = [ref.snp_update(int(loc)) for loc in locs]
    outs = [np.load(os.path.join(tmp_path, f"r{r}.npz")) for r in range(world)]
    for o in outs:
        assert list(o["its"]) == its                      # same convergence decision on every rank
        assert rel_err(o["lam"], ref.lambda_()) < 1e-11   # all-reduced sums: order differs only
        assert rel_err(o["gamma"], ref.gamma()) < 1e-10
    # every rank ends with bitwise the same replicated lambda (no broadcast needed)
    for o in outs[1:]:
        assert np.array_equal(o["lam"], outs[0]["lam"])


def test_in_launch_exchange_order_gives_every_rank_the_same_bits(tmp_path):
    """world = 2: group sums in member order, then all ranks' group sums in (rank, group) order (the exchange of
    ts_schedule on shards): same pass counts as one rank, lambda to rounding, and bitwise equal on both ranks."""
    n, l, k, seed, world = 2003, 24, 5, 43, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, l, k, seed, str(tmp_path), True), nprocs=world, join=True)
    y, _, _ = psd_genotypes(n, l, k, seed, 0.03)
    ref = op.Oracle(n, l, k)
    ref.load_bed_payload(pack_bed(y))
    ref.set_gamma(init_gamma(n, k, seed + 1))
    locs = np.random.default_rng(seed + 2).integers(0, l, size=12)
    its = [ref.snp_update(int(loc)) for loc in locs]
    outs = [np.load(os.path.join(tmp_path, f"r{r}.npz")) for r in range(world)]
    for o in outs:
        assert list(o["its"]) == its
        assert rel_err(o["lam"], ref.lambda_()) < 1e-11
        assert rel_err(o["gamma"], ref.gamma()) < 1e-10
    assert np.array_equal(outs[0]["lam"], outs[1]["lam"])
