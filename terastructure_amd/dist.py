"""Multi-GPU plumbing: one process per GPU, individuals sharded by
tsamd_shard_range, lambda_t all-reduced per pass by RCCL inside libtsamd.

torch.distributed is only the side channel here (unique-id broadcast, timing
barriers, gathering shard rows for output); any backend works (gloo on CPU).
"""
import os

import numpy as np


def init_process_group(backend="gloo"):
    """Rendezvous from the torchrun environment (RANK / WORLD_SIZE / MASTER_*)."""
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist, rank, world


def bootstrap_comm(engine, dist):
    """rank 0 makes the RCCL unique id, every rank joins the communicator.
    Raises on EVERY rank if any rank failed, so callers can fall back together."""
    err = None
    uid = [None]
    if dist.get_rank() == 0:
        try:
            uid = [engine.comm_unique_id()]
        except Exception as exc:  # noqa: BLE001
            err = exc
    dist.broadcast_object_list(uid, src=0)
    if uid[0] is not None:
        try:
            engine.comm_init(uid[0])
        except Exception as exc:  # noqa: BLE001
            err = exc
    elif err is None:
        err = RuntimeError("rank 0 could not create the RCCL unique id")
    if not all_ok(err is None, dist):
        raise RuntimeError(f"RCCL communicator unavailable: {err or 'a peer failed'}")


def all_ok(ok, dist):
    """True iff `ok` holds on every rank (one collective; call it unconditionally)."""
    import torch

    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return t.item() > 0


def bootstrap_p2p(engine, dist):
    """Map every rank's exchange buffer into every rank (tsamd_p2p_export / _connect).
    Raises on EVERY rank if any rank failed, so callers can fall back together."""
    err = None
    try:
        mine = engine.p2p_export()
    except Exception as exc:  # noqa: BLE001
        mine, err = b"", exc
    handles = [None] * dist.get_world_size()
    dist.all_gather_object(handles, mine)
    if err is None and all(len(h) == len(mine) and h for h in handles):
        try:
            engine.p2p_connect(handles)
        except Exception as exc:  # noqa: BLE001
            err = exc
    elif err is None:
        err = RuntimeError("a peer could not export its exchange buffer")
    if not all_ok(err is None, dist):
        raise RuntimeError(f"peer-to-peer exchange unavailable: {err or 'a peer failed'}")


def shard_bounds(n, world, shard_range):
    """[(begin, count)] for every rank."""
    return [shard_range(n, r, world) for r in range(world)]


def gather_rows(local, n, dist, shard_range):
    """Concatenate per-shard row blocks [count_r][k] into [n][k] on every rank."""
    import torch

    world = dist.get_world_size()
    bounds = shard_bounds(n, world, shard_range)
    k = local.shape[1]
    width = max(c for _, c in bounds)
    buf = torch.zeros((width, k), dtype=torch.float64)
    buf[:local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local))
    parts = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = np.empty((n, k), dtype=np.float64)
    for (b, c), part in zip(bounds, parts):
        out[b:b + c] = part[:c].numpy()
    return out


def sum_over_ranks(values, dist):
    """Sum a small vector of float64 over ranks (held-out log-likelihood, counts)."""
    import torch

    t = torch.tensor(list(values), dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()


def max_over_ranks(value, dist):
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
