"""bench.py's measurement legs (everything the JSON line carries besides `value`), world = 2 over gloo on CPU with engines
whose arithmetic is the oracle's: a leg that fails on ONE rank is recorded and costs only itself -- the roofline of the
timed kernel, the CPU baseline and the parity object still reach the line -- and no rank is left in a collective its peer
skipped.  (The real engines run the same function in tests/test_gpu_bench_contract.py.)"""
import argparse
import json
import os
import types

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import init_gamma, pack_bed, psd_genotypes
from test_distributed_cpu import ShardedOracle, _free_port


class _Cfg:
    max_inner = 10


class _FakeEngine:
    """one shard of a run whose arithmetic is the oracle's (ShardedOracle: all-reduce over gloo where libtsamd exchanges)"""

    def __init__(self, ts, n, l, k, rank, world, fail_leg, fail_rank):
        self.ts, self.n, self.l, self.k, self.rank, self.world = ts, n, l, k, rank, world
        self.fail = fail_leg if rank == fail_rank else None
        y, _, _ = psd_genotypes(n, l, k, 11, 0.02)
        self.payload = pack_bed(y)
        self.cfg = _Cfg()
        self.shard_begin, self.shard_count = ts.shard_range(n, rank, world)
        self.passes = 0
        self.mode = 2
        self.sh = None
        self.set_gamma(init_gamma(n, k, 5)[self.shard_begin:self.shard_begin + self.shard_count])

    # state
    def set_gamma(self, rows):
        full = init_gamma(self.n, self.k, 5)   # (the legs always restore the same start)
        if self.sh is not None:
            self.sh.orc.close()
        self.sh = ShardedOracle(self.n, self.l, self.k, self.payload, full, self.rank, self.world, self.ts.shard_range)

    def set_counts(self, c):
        assert not c.any()

    def set_lambda(self, j, lam):
        self.sh.orc.set_lambda(j, lam)

    def clear_pending(self):
        self.sh.pending = None

    def get_lambda(self, first=0, count=None):
        return self.sh.orc.lambda_()[first:first + (count or self.l)]

    def get_gamma(self):
        return self.sh.orc.gamma()[self.sh.b:self.sh.b + self.sh.c]

    def get_counts(self):
        return self.sh.orc.c_indiv()[self.sh.b:self.sh.b + self.sh.c]

    def download_bed(self, j):
        b, c = self.sh.b, self.sh.c
        return self.payload[j, b // 4:(b + c + 3) // 4]

    # the hot path
    def run_schedule(self, locs, hol_mode=0):
        for loc in locs:
            self.passes += self.sh.snp_update(int(loc), hol_mode)   # (collective inside: every rank gets here)

    def synchronize(self):
        pass

    def total_passes(self):
        return self.passes

    # measurement
    def launch_info(self):
        return {"kernels_per_snp": 0 if self.mode == 2 else 10}

    def set_launch_mode(self, mode):
        raise AssertionError("a sharded context must not be switched between launch modes for a measurement")

    def profile_enable(self, on):
        self.prof = on

    def profile_read(self):
        if self.fail == "roofline_timed_kernel":
            raise RuntimeError("tsamd error -4: scripted timeout")
        return dict(pass_launches=1, pass_ms=2.0, first_launches=0, first_ms=0.0)

    def schedule_geometry(self):
        return dict(workgroups=8, indivs_per_thread=1, exchange_levels=2, on_chip_per_thread=1)

    def probe_stream(self, reps):
        if self.fail == "stream_probe":
            raise RuntimeError("scripted probe failure")
        return 1.0, 2.0

    def set_heldout(self, loc, indivs):
        pass

    def heldout_eval(self, locs, run_updates=True):
        if run_updates:
            self.run_schedule(locs, 1)
        # (scripted on the call WITHOUT updates -- the leg's last one.  A real rank whose kernels time out leaves its peers'
        # kernels to their own bounded waits, so they fail too; the gloo all-reduce inside this stand-in's run_schedule has
        # no such bound, so the scripted failure must not leave the peer inside one)
        if self.fail == "validation_block" and not run_updates:
            raise RuntimeError("tsamd error -4: scripted timeout in the validation block")
        return -0.7 * 3 * len(locs), 3 * len(locs), None, None

    def holblock_info(self):
        return dict(batch=16, launches=1, locations=1)


def _legs_worker(rank, world, port, fail_leg, fail_rank, out_dir):
    import datetime
    import sys

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import terastructure_amd as real

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    n, l, k = 1003, 24, 5
    ts = types.SimpleNamespace(LAUNCH_PER_PASS=0, LAUNCH_PER_SNP=1, LAUNCH_PER_SCHEDULE=2, TsamdError=RuntimeError,
                               shard_range=real.shard_range)
    eng = _FakeEngine(ts, n, l, k, rank, world, fail_leg, fail_rank)
    args = argparse.Namespace(no_profile=False, steps=12, warmup=2, seed=3, validation_locs=4, cpu_seconds=0.4)
    locs = np.random.default_rng(1).integers(0, l, size=14).astype(np.uint32)
    m = bench.measure_legs(args, ts, eng, dist, rank, world, 0, n, l, k, eng.shard_count, locs, 2, True, lambda: init_gamma(n, k, 5))
    json.dump(m, open(os.path.join(out_dir, f"r{rank}.json"), "w"), default=str)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_leg,fail_rank", [(None, 0), ("validation_block", 1), ("roofline_timed_kernel", 0), ("stream_probe", 1)])
def test_one_failing_leg_costs_only_itself(tmp_path, fail_leg, fail_rank):
    world = 2
    mp.spawn(_legs_worker, args=(world, _free_port(), fail_leg, fail_rank, str(tmp_path)), nprocs=world, join=True)
    outs = [json.load(open(os.path.join(tmp_path, f"r{r}.json"))) for r in range(world)]
    assert outs[0]["legs"].keys() == outs[1]["legs"].keys()
    for name in outs[0]["legs"]:                                    # the same verdict on every rank
        assert outs[0]["legs"][name].split()[0].rstrip(":") == outs[1]["legs"][name].split()[0].rstrip(":"), name
    m, legs = outs[0], outs[0]["legs"]
    if fail_leg is not None:                                        # rank 0's line names the leg as failed, with the rank's own message or "on a peer rank"
        assert legs[fail_leg].startswith("failed") and outs[fail_rank]["legs"][fail_leg].startswith("failed: RuntimeError")
    if fail_leg != "roofline_timed_kernel":
        rf = m["roofline"]                                          # the timed kernel's roofline: from the timed mode alone
        assert rf is not None and rf["bound"] == "fp64_valu" and rf["avg_launch_us"] == 2000.0 and rf["launches_timed"] == 1
        assert rf["updates_per_launch"] == 12 and rf["launch_per_snp"] is None and "algorithmic" in rf["flops_source"] and rf["executed"]["flops_per_update"] is None
        assert legs["roofline_timed_kernel"] == "ok"
    else:
        assert m["roofline"] is None and legs["roofline_timed_kernel"].startswith("failed")
    # the CPU baseline (rank 0's oracle on the gathered columns) is reported whatever the GPU legs did
    cb = m["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] == 2 and legs["cpu_baseline"] == "ok"
    if fail_leg in (None, "stream_probe"):
        pv = m["parity"]                                            # both shards' states gathered and compared on rank 0
        assert pv["ok"] and pv["c_n_equal"] and pv["lambda_rel_err"] < 1e-9 and pv["gamma_rel_err"] < 1e-9
        assert "other_launch_modes" not in pv and outs[1]["parity"] is None
    if fail_leg is None:
        vb = m["validation_block"]
        assert vb["locations"] == 4 and vb["heldout_entries"] == 2 * 3 * 4 and abs(vb["mean_loglik"] + 0.7) < 1e-9
        assert all(v == "ok" for name, v in legs.items() if name != "device_copy")   # (no device here: that leg fails, alone)
        assert legs["device_copy"].startswith("failed") and m["roofline"]["device_copy_GBps"] is None
    if fail_leg == "validation_block":
        assert m["validation_block"] is None and legs["validation_block"].startswith("failed")
        assert m["parity"]["ok"]                                    # (it ran before the failing leg)
    if fail_leg == "roofline_timed_kernel":                         # a sharded context after a failed kernel leg: no further kernels
        assert legs["parity_vs_cpu_baseline"].startswith("skipped") and legs["validation_block"].startswith("skipped")
    if fail_leg == "stream_probe":
        assert m["roofline"]["probe_read_us"] is None
