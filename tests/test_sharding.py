"""CPU: the N>1 path — contiguous image shards + one all-gather of per-image records — over gloo, world_size 2."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from facet_amd.sharding import shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 100000, 256 * 8 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    import torch.distributed as dist
    from facet_amd.sharding import shard_range, gather_scores, gather_ragged
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # equal shards: per-image record = (global index, index^2)
        lo, hi = shard_range(8, world, rank)
        rec = np.stack([np.arange(lo, hi), np.arange(lo, hi) ** 2], 1).astype(np.float32)
        full = gather_scores(rec, world)
        ok1 = full.shape == (8, 2) and np.array_equal(full[:, 0], np.arange(8)) and np.array_equal(full[:, 1], np.arange(8) ** 2)
        # ragged shards
        lo, hi = shard_range(n_items, world, rank)
        local = np.arange(lo, hi, dtype=np.float32) * 0.5
        allv = gather_ragged(local, n_items, world, rank)
        ok2 = np.array_equal(allv, np.arange(n_items, dtype=np.float32) * 0.5)
        q.put((rank, bool(ok1), bool(ok2)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 10])
def test_allgather_world2_gloo(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True, True), (1, True, True)]


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` typed directly (what the driver's scaling run does): the parent starts two rank processes itself,
    relays rank 0's line and returns their status. --dry-run keeps the engine out so the launch / rendezvous / gather path runs on CPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"dry_run": True, "n_gpus": 2, "gathered_ok": True}
    # a failing child makes the parent fail
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--workload", "nope"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_committed_bench_line_has_the_contract_fields():
    """The driver parses ONE JSON line from `python bench.py`: the line committed under profiles/ (the end-of-round run) must carry
    every field of the contract, incl. the roofline and cpu_baseline objects, and its numbers must be consistent with each other."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    logs = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench_default.json.log")))
    assert logs, "no committed bench line"
    line = [l for l in open(logs[-1]) if l.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["value"] > 0
    # value = images of the timed region / its time
    images = d["config"]["global_batch"] * d["steps"]
    assert abs(d["value"] - images / (d["ms_per_step"] * d["steps"] * 1e-3)) < 0.02 * d["value"]
    for name, sub in d.get("sub", {}).items():
        assert sub["value"] > 0 and "roofline" in sub and "workload" in sub["config"], name
