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


def gather_device(local_t, world):
    """All-gather of a torch float32 tensor [n, R] that already lives where the collective runs: on the GPU under "nccl" (= RCCL
    over xGMI; the engine wrote the records straight into this buffer with Engine.ensemble_score_dev, so nothing crosses PCIe before
    the exchange), on the CPU under "gloo". Returns the [world * n, R] tensor on the same device; the caller copies it to the host
    once, after the step."""
    if world == 1:
        return local_t
    import torch
    import torch.distributed as dist
    if dist.get_backend() != "nccl" and local_t.is_cuda:       # gloo rehearsal of the N > 1 control flow on one GPU
        local_t = local_t.cpu()
    out = torch.empty((world * local_t.shape[0],) + tuple(local_t.shape[1:]), dtype=local_t.dtype, device=local_t.device)
    dist.all_gather_into_tensor(out, local_t.contiguous())
    return out


_face_pool = None


def _face_worker():
    global _face_pool
    if _face_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _face_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="facet-faces")
    return _face_pool


def score_shard(engine, images, n_items, world, rank, faces=None, record_floats=789, face_engine=None):
    """One multi-GPU step of the ensemble for this rank's contiguous block of a global batch of n_items images: `images` is the
    rank's (device_ptr, n, h, w) block (n = hi - lo of shard_range). Records are produced in a device buffer padded to the largest
    shard, gathered once, and returned as a host array [n_items, R] in global image order (identical on every rank).
    faces: None, or (det_size, det_thresh, nms_thresh, max_faces) to append [count, max_faces x 739 face slots] per image.
    face_engine: a second Engine (own context and stream on the same device) holding the face graphs; the face stage, whose NMS /
    alignment glue runs on the host between launches, then runs in a worker thread beside the ensemble instead of after it (the same
    arrangement as BatchScorer's aux_engine). Results are identical either way."""
    import torch
    from ._lib import FE_FACE_FLOATS
    lo, hi = shard_range(n_items, world, rank)
    n = hi - lo
    cap = -(-n_items // world)
    R = record_floats + (1 + faces[3] * FE_FACE_FLOATS if faces else 0)
    dev = torch.device("cuda", engine.device)
    rec = torch.zeros((cap, R), dtype=torch.float32, device=dev)
    mask = 0
    if n:
        assert images[1] == n, (images[1], lo, hi)
        fut = None
        if faces and face_engine is not None and face_engine is not engine:
            fut = _face_worker().submit(face_engine.face_analyze, images, faces[0], faces[1], faces[2], faces[3])
        try:
            mask = engine.ensemble_score_dev(images, rec.data_ptr(), R)
        except BaseException:
            if fut is not None:
                fut.exception()            # never leave the worker running against buffers the caller may free
            raise
        if faces:
            f, counts, _ = fut.result() if fut is not None else engine.face_analyze(images, faces[0], faces[1], faces[2], faces[3])
            extra = np.concatenate([counts[:, None].astype(np.float32), f.reshape(n, -1)], axis=1)
            rec[:n, record_floats:] = torch.from_numpy(extra).to(dev)      # the face glue runs on the host: a few KB per image go up
    full = gather_device(rec, world).cpu().numpy()
    if world == 1:
        return full[:n], mask
    parts = []
    for r in range(world):
        a, b = shard_range(n_items, world, r)
        parts.append(full[r * cap: r * cap + (b - a)])
    return np.concatenate(parts, 0), mask
