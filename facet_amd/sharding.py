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


class ShardBuffers:
    """Per-shard buffers of score_shard, allocated ONCE for a (device, rows, record width) and reused by every step: the device record
    table the engine writes and the collective reads, the gathered table, and pinned host staging for the face slots (up) and the
    gathered table (down).

    Stream contract. fe_ensemble_score_dev writes `rec` from the ENGINE's stream (created hipStreamNonBlocking: nothing orders it
    against torch's streams) and returns only after that stream has drained. The zero fill below runs on torch's current stream, so
    the constructor synchronises that stream before the buffer is ever handed to the engine; afterwards every torch access to `rec`
    (face-slot upload, collective, copy down) is queued after an engine call that has already completed on the host, and the next
    engine call starts only after the blocking copy down of the previous step - there is no point where two streams touch the table
    without a host-side ordering between them."""

    def __init__(self, device, cap, R, world):
        import torch
        self.key = (device, cap, R, world)
        dev = torch.device("cuda", device)
        self.rec = torch.zeros((cap, R), dtype=torch.float32, device=dev)
        self.full = self.rec if world == 1 else torch.empty((world * cap, R), dtype=torch.float32, device=dev)
        self.host = torch.empty((world * cap, R), dtype=torch.float32).pin_memory()
        self.face_host = None
        torch.cuda.current_stream(dev).synchronize()      # the fill has landed before the engine's stream may write the table


_shard_buffers = {}


def shard_buffers(device, cap, R, world):
    key = (device, cap, R, world)
    b = _shard_buffers.get(key)
    if b is None:
        if len(_shard_buffers) >= 4:      # a handful of shapes per process at most (primary workload + sub-measurements)
            _shard_buffers.pop(next(iter(_shard_buffers)))
        b = _shard_buffers[key] = ShardBuffers(device, cap, R, world)
    return b


def models_that_run(engine):
    """The models_run bitmask fe_ensemble_score would report on this engine (1 topiq | 2 clip | 4 samp): selection AND loaded
    models. A property of the context, not of whether this rank had images to score."""
    from ._lib import FE_MODEL_TOPIQ, FE_MODEL_CLIP, FE_MODEL_SAMP, FE_MODEL_U2NETP
    sel = getattr(engine, "ensemble_mask", 7)
    loaded = lambda m: engine.model_precision(m) is not None
    return ((1 if sel & 1 and loaded(FE_MODEL_TOPIQ) else 0) | (2 if sel & 2 and loaded(FE_MODEL_CLIP) else 0) |
            (4 if sel & 4 and loaded(FE_MODEL_SAMP) and loaded(FE_MODEL_U2NETP) else 0))


def score_shard(engine, images, n_items, world, rank, faces=None, record_floats=789, face_engine=None):
    """One multi-GPU step of the ensemble for this rank's contiguous block of a global batch of n_items images: `images` is the
    rank's (device_ptr, n, h, w) block (n = hi - lo of shard_range; n may be 0 when n_items < world). Records are produced in a
    device buffer padded to the largest shard, gathered once, and returned as a host array [n_items, R] in global image order;
    array and mask are identical on every rank (the mask is what the context would run - see models_that_run - so a rank with an
    empty shard reports the same value as its peers).
    faces: None, or (det_size, det_thresh, nms_thresh, max_faces) to append [count, max_faces x 739 face slots] per image.
    face_engine: a second Engine (own context and stream on the same device) holding the face graphs; the face stage, whose NMS /
    alignment glue runs on the host between launches, then runs in a worker thread beside the ensemble instead of after it (the same
    arrangement as BatchScorer's aux_engine). Results are identical either way.
    Buffers: ShardBuffers (allocated once per shape; its docstring states the stream contract)."""
    import torch
    from ._lib import FE_FACE_FLOATS
    lo, hi = shard_range(n_items, world, rank)
    n = hi - lo
    cap = -(-n_items // world)
    R = record_floats + (1 + faces[3] * FE_FACE_FLOATS if faces else 0)
    buf = shard_buffers(engine.device, cap, R, world)
    rec = buf.rec
    mask = models_that_run(engine)      # what a rank without images reports; ranks that ran report the engine's own value
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
            if buf.face_host is None:
                buf.face_host = torch.empty((cap, R - record_floats), dtype=torch.float32).pin_memory()
            fh = buf.face_host.numpy()
            fh[:n, 0] = counts
            fh[:n, 1:] = f.reshape(n, -1)
            # the face glue runs on the host: a few KB per image go up, from pinned staging, queued behind nothing but earlier steps
            rec[:n, record_floats:].copy_(buf.face_host[:n], non_blocking=True)
    if n < cap and (faces or n == 0):
        rec[n:].zero_()                    # padding rows of a ragged / empty shard (torch stream; the engine is idle here)
    if world == 1:
        gathered = rec
    else:
        import torch.distributed as dist
        if dist.get_backend() != "nccl":       # gloo rehearsal of the N > 1 control flow on one GPU: the collective runs on host tensors
            gathered = gather_device(rec, world)
        else:
            dist.all_gather_into_tensor(buf.full, rec)
            gathered = buf.full
    if gathered.is_cuda:
        buf.host[: gathered.shape[0]].copy_(gathered, non_blocking=False)      # blocking: returns when the table is on the host
        full = buf.host[: gathered.shape[0]].numpy()
    else:
        full = gathered.numpy()
    if world == 1:
        return full[:n].copy(), mask
    parts = []
    for r in range(world):
        a, b = shard_range(n_items, world, r)
        parts.append(full[r * cap: r * cap + (b - a)])
    return np.concatenate(parts, 0), mask
