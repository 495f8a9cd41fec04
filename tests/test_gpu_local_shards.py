"""Several shards driven by ONE process (tsamd_p2p_connect_local, tsamd_run_schedule_all): the
way the drop-in host uses all GPUs of a node from its main thread.  On the one-GPU test box all
shards share device 0, which exercises everything except the xGMI hop -- and needs one hardware
queue per shard, hence a fresh process with GPU_MAX_HW_QUEUES raised (local_shards_worker.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.spawns]
HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(*args):
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "local_shards_worker.py"), *map(str, args)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "worker ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("world,n,k", [(2, 3001, 5), (3, 20000, 8), (4, 5000, 20)])
def test_local_shards_match_oracle(world, n, k):
    _worker("match", world, n, k)


def test_local_shards_deep_queue():
    """Far more kernels than a device queue holds (pass cap 100 as in -compute-beta, 1500
    locations): run_schedule_all interleaves the shards in bounded batches, so the one driving
    thread never blocks with a peer's work unsubmitted."""
    _worker("deep")


def test_local_connect_errors():
    _worker("errors")
