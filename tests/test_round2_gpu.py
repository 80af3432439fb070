"""GPU: round-2 additions - selectable GatedConv activations, pyramid sizes that are not multiples of 32, gather-buffer compaction
with a micro-batch larger than the tower chunk, device-resident ensemble records, and the engine-level 2-rank run."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from facet_amd._lib import (FE_MODEL_TOPIQ, FE_MODEL_CLIP, FE_MODEL_AESTHETIC, FE_MODEL_SAMP, FE_MODEL_U2NETP, FE_RECORD_FLOATS)
from facet_amd.weights import synthetic_state_dict, synthetic_images

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gate,wblk", [("gelu", "gelu"), ("softplus", "relu"), ("relu", "softplus"), ("softplus", "gelu")])
def test_topiq_gate_activation_variants_match_oracle(engine, gate, wblk):
    """pyiqa's GatedConv activations are a load-time option (fe_topiq_configure): every variant must equal the oracle built with the
    same choice, and the variants must differ from each other (so the option is really wired through)."""
    from oracle.topiq import CFANet
    sd = synthetic_state_dict("topiq", seed=21)
    imgs = synthetic_images(4, 2, 160, 192)
    try:
        engine.topiq_configure(gate, wblk)
        engine.load_weights(FE_MODEL_TOPIQ, sd)
        got = engine.topiq_score(imgs)
    finally:
        engine.topiq_configure("gelu", "gelu")
    net = CFANet(gate_act=gate, weight_blk_act=wblk).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    with torch.no_grad():
        x = torch.from_numpy(imgs.astype(np.float32) / 255.0).permute(0, 3, 1, 2)
        ref = net(x).flatten().numpy()
        base = CFANet().eval()
        base.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        ref_default = base(x).flatten().numpy()
    assert (np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)).max() < 1e-3, (got, ref)
    if (gate, wblk) != ("gelu", "gelu"):
        assert np.abs(ref - ref_default).max() > 1e-4          # a different function, not the default in disguise


def test_softplus_epilogue_matches_torch(engine):
    rng = np.random.default_rng(3)
    x = rng.normal(0, 8, (1, 32, 6, 40)).astype(np.float32)     # includes values beyond torch's threshold of 20 after scaling
    w = rng.normal(0, 0.5, (48, 32, 1, 1)).astype(np.float32)
    got = engine.conv2d(x, w, act="softplus")
    ref = torch.nn.functional.softplus(torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w))).numpy()
    assert ref.max() > 20 and np.abs(got - ref).max() < 2e-4 * np.abs(ref).max()


@pytest.mark.parametrize("hw", [(40, 72), (33, 95), (1100, 36)])
def test_pyramid_shapes_for_sizes_not_multiple_of_32(engine, hw):
    """Engine.topiq_features sizes its output from the engine's own conv / pool arithmetic (fe_topiq_feature_shape), also when the
    long edge exceeds 1024 (LANCZOS cap first)."""
    from oracle.resnet import ResNet50Features
    from PIL import Image
    sd = synthetic_state_dict("topiq", seed=3)
    engine.load_weights(FE_MODEL_TOPIQ, sd)
    imgs = synthetic_images(6, 1, *hw)
    net = ResNet50Features().eval()
    net.load_state_dict({k[len("semantic_model."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("semantic_model.")})
    small = imgs
    if max(hw) > 1024:
        s = 1024 / max(hw)
        small = np.asarray(Image.fromarray(imgs[0]).resize((int(hw[1] * s), int(hw[0] * s)), Image.LANCZOS))[None]
    m, sdv = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    with torch.no_grad():
        ref = net((torch.from_numpy(small.astype(np.float32) / 255.0).permute(0, 3, 1, 2) - m) / sdv)
    for level in (0, 2, 4):
        got = engine.topiq_features(imgs, level)
        r = ref[level].numpy()
        assert got.shape == r.shape == (1,) + engine.topiq_feature_shape(hw[0], hw[1], level)
        assert np.abs(got - r).max() < 1e-3 * np.abs(r).max()


def test_gather_buffer_compaction_with_microbatch_larger_than_chunk():
    """fe_set_microbatch accepts up to 256 while the tower chunks are at most 128: 300 images pushed 256 at a time leave more
    crops behind a chunk than the chunk holds, so the move to the front of the gather buffer has to be cut into disjoint pieces."""
    from facet_amd import Engine
    e = Engine(0, arena_bytes=24 << 30)
    for mid, name in ((FE_MODEL_CLIP, "clip"), (FE_MODEL_AESTHETIC, "aesthetic"), (FE_MODEL_U2NETP, "u2netp"), (FE_MODEL_SAMP, "samp_net")):
        e.load_weights(mid, synthetic_state_dict(name, 5))
    imgs = synthetic_images(12, 300, 64, 64)
    e.set_microbatch(32)
    ref, mask = e.ensemble_score(imgs)
    e.set_microbatch(256)
    got, mask2 = e.ensemble_score(imgs)
    e.close()
    assert mask == mask2 == 6
    assert np.abs(got[:, 21:] - ref[:, 21:]).max() <= 2e-5 and np.abs(got[:, 1:21] - ref[:, 1:21]).max() <= 2e-5 * max(1.0, np.abs(ref[:, 1:21]).max())


def test_device_records_equal_host_records(engine):
    for mid, name in ((FE_MODEL_TOPIQ, "topiq"), (FE_MODEL_CLIP, "clip"), (FE_MODEL_AESTHETIC, "aesthetic"), (FE_MODEL_U2NETP, "u2netp"),
                      (FE_MODEL_SAMP, "samp_net")):
        engine.load_weights(mid, synthetic_state_dict(name, 13))
    imgs = synthetic_images(8, 5, 96, 128)
    engine.set_microbatch(2)
    host, mask = engine.ensemble_score(imgs)
    ld = FE_RECORD_FLOATS + 11
    buf = engine.dev_alloc(5 * ld * 4)
    sentinel = np.full((5, ld), -7.0, np.float32)
    engine.h2d(buf, sentinel)
    mask2 = engine.ensemble_score_dev(imgs, buf, ld)
    out = np.empty((5, ld), np.float32)
    engine.d2h(out, buf)
    engine.dev_free(buf)
    assert mask == mask2 == 7
    assert np.array_equal(out[:, :FE_RECORD_FLOATS], host) and (out[:, FE_RECORD_FLOATS:] == -7.0).all()
    # model selection: only the selected models run, the other fields stay 0
    engine.ensemble_select(3)
    sel, m3 = engine.ensemble_score(imgs)
    engine.ensemble_select(7)
    assert m3 == 3 and np.array_equal(sel[:, 0], host[:, 0]) and (sel[:, 2:21] == 0).all() and np.array_equal(sel[:, 21:], host[:, 21:])


_RANK_SCRIPT = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, os.environ["FACET_ROOT"])
os.environ["FACET_AMD_SYNTHETIC"] = "1"
import torch, torch.distributed as dist
from facet_amd import Engine
from standins import synthetic_onnx as SO
from facet_amd._lib import (FE_MODEL_TOPIQ, FE_MODEL_CLIP, FE_MODEL_AESTHETIC, FE_MODEL_SAMP, FE_MODEL_U2NETP, FE_GRAPH_FACE_DET,
                            FE_GRAPH_FACE_LMK, FE_GRAPH_FACE_REC)
from facet_amd.weights import synthetic_state_dict, synthetic_images
from facet_amd.sharding import shard_range, score_shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
N, H, W = int(os.environ["FACET_N"]), 160, 192
backend = os.environ.get("FACET_BACKEND", "gloo")
DEV = rank if backend == "nccl" else 0       # gloo: two ranks share the one GPU of the test box; nccl (= RCCL): one GPU per rank
torch.cuda.set_device(DEV)
if world > 1:
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", DEV))
    else:
        dist.init_process_group("gloo")
side = torch.cuda.Stream(DEV) if os.environ.get("FACET_SIDE_STREAM") else None      # score_shard under a non-default torch stream
eng = Engine(DEV, arena_bytes=6 << 30, precision=os.environ.get("FACET_PREC", "f32"))
for mid, name in ((FE_MODEL_TOPIQ, "topiq"), (FE_MODEL_CLIP, "clip"), (FE_MODEL_AESTHETIC, "aesthetic"), (FE_MODEL_U2NETP, "u2netp"),
                  (FE_MODEL_SAMP, "samp_net")):
    eng.load_weights(mid, synthetic_state_dict(name, 13))
# the sharded ranks keep the face graphs on a second context that runs beside the ensemble (score_shard face_engine, what bench.py
# does); the single rank runs them on the one context, after the ensemble: the rows must not depend on that
feng = Engine(DEV, arena_bytes=2 << 30) if world > 1 else eng
feng.graph_load(FE_GRAPH_FACE_DET, SO.scrfd_like(seed=12, size=160)[0])
feng.graph_load(FE_GRAPH_FACE_LMK, SO.landmark_like(seed=13)[0])
feng.graph_load(FE_GRAPH_FACE_REC, SO.arcface_iresnet(layers=(1, 1, 1, 1), seed=14)[0])
eng.set_microbatch(2); feng.set_microbatch(2)
imgs = synthetic_images(31, N, H, W)          # the GLOBAL batch; every rank uploads only its contiguous block
lo, hi = shard_range(N, world, rank)
d = eng.dev_alloc(max(1, hi - lo) * H * W * 3)
if hi > lo:
    eng.h2d(d, imgs[lo:hi])
with (torch.cuda.stream(side) if side is not None else torch.cuda.device(DEV)):
    for _ in range(2 if side is not None else 1):      # twice under the side stream: the second step reuses the cached ShardBuffers
        rec, mask = score_shard(eng, (d, hi - lo, H, W), N, world, rank, faces=((160, 160), 0.3, 0.4, 2), face_engine=feng)
np.save(os.path.join(os.environ["FACET_OUT"], f"rec_w{world}_r{rank}.npy"), rec)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
eng.dev_free(d)
if feng is not eng:
    feng.close()
eng.close()
print(json.dumps({"rank": rank, "shape": list(rec.shape), "mask": mask}))
'''


def _gpu_count():
    import torch
    return torch.cuda.device_count()      # counting devices does not initialise the GPU in this process


@pytest.mark.parametrize("n_items,prec,backend", [(6, "f32", "gloo"), (5, "f32", "gloo"), (5, "bf16", "gloo"), (1, "f32", "gloo"),
                                                   (5, "f32", "nccl")])
def test_two_ranks_score_their_shards_and_gather_the_single_rank_result(tmp_path, n_items, prec, backend):
    """Engine-level N > 1 run: two fresh processes (gloo, both on GPU 0) each score their shard_range block through
    fe_ensemble_score_dev + fe_face_analyze, all-gather the fixed-size records (789 + 1 + 2 x 739 floats) and must end up with
    exactly the rows a single rank computes for the whole batch - also for a ragged N (5 = 3 + 2), and with the models committed in
    bf16 (BASELINE configs[3]: the 16gb profile in bf16 sharded across ranks; the face graphs stay fp32). n_items = 1: rank 1's shard is
    empty - it still joins the collective and reports the same mask. backend nccl: the RCCL branch proper (one GPU per rank, records
    gathered device to device); needs two visible GPUs and is skipped on a one-GPU box."""
    if backend == "nccl" and _gpu_count() < 2:
        pytest.skip(f"the RCCL (nccl) branch needs 2 visible GPUs, this box has {_gpu_count()}")
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    base.update(FACET_ROOT=ROOT, FACET_OUT=str(tmp_path), FACET_N=str(n_items), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                HSA_ENABLE_IPC_MODE_LEGACY="0", FACET_PREC=prec, FACET_BACKEND=backend)
    one = subprocess.run([sys.executable, str(script)], env=dict(base, RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(base, RANK=str(r), WORLD_SIZE="2"), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    single = np.load(tmp_path / "rec_w1_r0.npy")
    r0, r1 = np.load(tmp_path / "rec_w2_r0.npy"), np.load(tmp_path / "rec_w2_r1.npy")
    assert single.shape == (n_items, FE_RECORD_FLOATS + 1 + 2 * 739)
    assert np.array_equal(r0, r1)                       # every rank holds the same gathered table
    # row for row the single-rank result. The towers batch their crops per call, so a shard sees other chunk boundaries than the
    # whole batch: identical up to fp32 summation order inside the GEMM tiles, not bit-identical
    # (bf16: a crop that moves to another position of a tower batch meets other tile boundaries and rounds differently - 8 bits)
    tol = 5e-5 if prec == "f32" else 2e-2
    assert np.abs(r0 - single).max() <= tol * max(1.0, np.abs(single).max()), float(np.abs(r0 - single).max())
    assert np.array_equal(r0[:, FE_RECORD_FLOATS], single[:, FE_RECORD_FLOATS])      # face counts
    assert json.loads(one.stdout.strip().splitlines()[-1])["mask"] == 7
    for so, _ in outs:                                   # also the rank whose shard is empty (n_items < world)
        assert json.loads(so.strip().splitlines()[-1])["mask"] == 7


def test_score_shard_under_a_non_default_torch_stream(tmp_path):
    """the record table is filled by torch (current stream) and written by the engine (its own non-blocking stream). One
    rank scores the batch with a side stream current - twice, so the second step runs on the cached buffers - and must reproduce the
    default-stream rows exactly (ShardBuffers' docstring states the ordering that guarantees it)."""
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    base.update(FACET_ROOT=ROOT, FACET_N="3", HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", WORLD_SIZE="1")
    rows = []
    for tag, extra in (("plain", {}), ("side", {"FACET_SIDE_STREAM": "1"})):
        out = tmp_path / tag
        out.mkdir()
        r = subprocess.run([sys.executable, str(script)], env=dict(base, FACET_OUT=str(out), **extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        rows.append(np.load(out / "rec_w1_r0.npy"))
    assert rows[0].shape[0] == 3 and np.abs(rows[0][:, 0]).min() > 0
    assert np.array_equal(rows[0], rows[1])
