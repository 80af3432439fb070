"""Image sharding across GPUs: one process per GPU, contiguous index blocks, one all-gather of records.

Net-new relative to the reference (single process, device 0 hard-coded: models/model_manager.py:613,
analyzers/face.py:38); every image is scored independently (no cross-image state in
processing/batch_processor.py:169-360), so the only exchange is the all-gather of fixed-size per-image
result records. Backend: torch.distributed "nccl" (= RCCL over xGMI) on GPUs, "gloo" in CPU tests.
"""
import numpy as np


def shard_range(n_items, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; the first n_items % world ranks get one extra item."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} / world {world}")
    q, r = divmod(int(n_items), world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_scores(local, world, local_rank=0):
    """All ranks receive the concatenation (rank order) of every rank's `local` array (same shape on all
    ranks; pad ragged shards with gather_ragged). local: float32 [n] or [n, R]."""
    local = np.ascontiguousarray(local, dtype=np.float32)
    if world == 1:
        return local
    import torch
    import torch.distributed as dist
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    t = torch.from_numpy(local).to(dev)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, t)
    return out.cpu().numpy()


def gather_ragged(local, n_items, world, rank, local_rank=0):
    """All-gather for shard_range-partitioned records when n_items % world != 0: shards are padded to the
    largest shard, gathered once, and the padding dropped."""
    local = np.ascontiguousarray(local, dtype=np.float32)
    lo, hi = shard_range(n_items, world, rank)
    assert local.shape[0] == hi - lo, (local.shape, lo, hi)
    if world == 1:
        return local
    cap = -(-n_items // world)
    pad = np.zeros((cap,) + local.shape[1:], np.float32)
    pad[: hi - lo] = local
    full = gather_scores(pad, world, local_rank)
    parts = []
    for r in range(world):
        a, b = shard_range(n_items, world, r)
        parts.append(full[r * cap: r * cap + (b - a)])
    return np.concatenate(parts, 0)
