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


# ---- bench.py's exchange self-test (N > 1): its control flow on CPU ---------------------------------------------
class _FakeTs:
    """Stand-in for the terastructure_amd module: engines whose arithmetic is the oracle's (ShardedOracle above) and whose
    failures are scripted, so that choose_exchange's collective sequence can be exercised without a GPU."""
    LAUNCH_PER_PASS = 0

    def __init__(self, shard_range, gamma_full, fail_mode, fail_rank):
        self.shard_range, self.gamma_full, self.fail_mode, self.fail_rank = shard_range, gamma_full, fail_mode, fail_rank
        self.created = 0
        ts = self

        class Engine:
            def __init__(self, n, l, k, device=0, rank=0, world=1):
                ts.created += 1
                self.n, self.l, self.k, self.rank, self.world = n, l, k, rank, world
                # candidates are created in the order rccl, p2p, p2p_schedule, p2p_schedule3
                self.mode = ["rccl", "p2p", "p2p_schedule", "p2p_schedule3"][ts.created - 1]
                self.fail_now = False

            def synth_genotypes(self, theta_shard, beta, seed=1):
                y, _, _ = psd_genotypes(self.n, self.l, self.k, seed, 0.02)
                self.payload = pack_bed(y)

            def set_gamma(self, rows):
                self.sh = ShardedOracle(self.n, self.l, self.k, self.payload, ts.gamma_full, self.rank, self.world, ts.shard_range)

            def comm_unique_id(self):
                raise RuntimeError("no RCCL on the CPU box")

            def comm_init(self, uid):
                raise RuntimeError("no RCCL on the CPU box")

            def p2p_export(self):
                return b"h" * 64

            def p2p_connect(self, handles):
                assert len(handles) == self.world

            def launch_info(self):
                return {"kernels_per_snp": 0}

            def set_launch_mode(self, mode):
                pass

            def download_bed(self, j):
                b, c = self.sh.b, self.sh.c
                return self.payload[j, b // 4:(b + c + 3) // 4]

            def run_schedule(self, locs, hol_mode=0):
                for loc in locs:
                    self.sh.snp_update(int(loc), hol_mode)   # (collective inside: every rank gets here)
                self.fail_now = self.mode == ts.fail_mode and self.rank == ts.fail_rank

            def synchronize(self):
                if self.fail_now:                             # ... and ONE rank reports a failure afterwards, like a
                    raise RuntimeError("tsamd error -4: scripted timeout")   # bounded in-kernel wait that gave up

            def get_lambda(self):
                return self.sh.orc.lambda_()

            def get_gamma(self):
                return self.sh.orc.gamma()[self.sh.b:self.sh.b + self.sh.c]

            def get_counts(self):
                return self.sh.orc.c_indiv()[self.sh.b:self.sh.b + self.sh.c]

            def close(self):
                self.sh.orc.close()

        self.Engine = Engine


def _selftest_worker(rank, world, port, n, k, fail_mode, fail_rank, out_dir):
    import datetime
    import sys

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import terastructure_amd as ts

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    gamma_full = init_gamma(n, k, 5)
    fake = _FakeTs(ts.shard_range, gamma_full, fail_mode, fail_rank)
    fake.shard_range = ts.shard_range
    sb, sc = ts.shard_range(n, rank, world)
    chosen, report = bench.choose_exchange(fake, dist, rank, world, 0, n, k, None, gamma_full, (sb, sc), 2)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([str(chosen), str(sorted(report["valid"].items()))]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_mode,fail_rank", [("p2p", 1), ("p2p_schedule", 0), (None, 0)])
def test_exchange_selftest_keeps_ranks_aligned_when_one_rank_fails(tmp_path, fail_mode, fail_rank):
    """bench.py's start-up self-test of the exchanges (N > 1), world = 2 on CPU with engines whose arithmetic is the
    oracle's: a candidate whose run fails on ONE rank only is marked invalid on every rank and nobody is left waiting in a
    collective the failing rank skipped (a 4-rank rehearsal once hung for gloo's 30 minutes that way); the remaining
    candidates are checked against the oracle and one of them is chosen -- the same one on every rank."""
    world, n, k = 2, 1003, 5
    port = _free_port()
    mp.spawn(_selftest_worker, args=(world, port, n, k, fail_mode, fail_rank, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(os.path.join(tmp_path, f"r{r}.npy")) for r in range(world)]
    assert list(outs[0]) == list(outs[1])                      # same verdict everywhere
    chosen, valid = outs[0][0], dict(eval(outs[0][1]))
    assert valid["rccl"] is False
    for m in ("p2p", "p2p_schedule", "p2p_schedule3"):
        assert valid[m] is (m != fail_mode), (m, valid)
    assert chosen in [m for m in ("p2p", "p2p_schedule", "p2p_schedule3") if m != fail_mode]
