"""BASELINE.json's image size (1024x1024) through size-independent properties, plus one oracle comparison at full size.

The parity tests elsewhere run at sizes the CPU oracle finishes quickly; here the same entry points run at the benchmark's
resolution with several micro-batches in flight: batch / micro-batch invariance, permutation equivariance, determinism, and
agreement between the single-model call and the ensemble record. One image is also scored by the CPU oracle (a few seconds).
"""
import numpy as np
import pytest
import torch

from facet_amd._lib import FE_MODEL_TOPIQ
from facet_amd.weights import synthetic_state_dict, synthetic_images

from conftest import assert_int_boxes_match

pytestmark = pytest.mark.gpu
HW = 1024


@pytest.fixture(scope="module")
def big_engine():
    from facet_amd import Engine
    e = Engine(0, arena_bytes=24 << 30)
    e.load_weights(FE_MODEL_TOPIQ, synthetic_state_dict("topiq", seed=3))
    yield e
    e.close()


def test_topiq_fullsize_properties(big_engine):
    e = big_engine
    imgs = synthetic_images(2, 20, HW, HW)
    e.set_microbatch(8)
    a = e.topiq_score(imgs)                       # 3 micro-batches (8, 8, 4)
    assert a.shape == (20,) and np.isfinite(a).all() and a.std() > 0
    assert np.array_equal(a, e.topiq_score(imgs))                                      # deterministic
    e.set_microbatch(5)
    b = e.topiq_score(imgs)                       # different micro-batching: same per-image results
    assert np.abs(a - b).max() <= 1e-5 * np.abs(a).max()
    one = np.concatenate([e.topiq_score(imgs[i:i + 1]) for i in (0, 7, 19)])            # alone == inside a batch
    assert np.abs(one - a[[0, 7, 19]]).max() <= 1e-5 * np.abs(a).max()
    perm = np.random.default_rng(0).permutation(20)
    assert np.abs(e.topiq_score(imgs[perm]) - a[perm]).max() <= 1e-5 * np.abs(a).max()  # permutation equivariance
    d = e.dev_alloc(imgs.nbytes)
    e.h2d(d, imgs)
    assert np.array_equal(e.topiq_score((d, 20, HW, HW)), b)                            # resident input == host input
    e.dev_free(d)
    rec, mask = e.ensemble_score(imgs[:6])
    assert mask == 1 and np.abs(rec[:, 0] - a[:6]).max() <= 1e-5 * np.abs(a).max() and not rec[:, 1:].any()


def test_topiq_fullsize_vs_oracle(big_engine):
    from oracle.topiq import CFANet
    sd = synthetic_state_dict("topiq", seed=3)
    net = CFANet().eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    img = synthetic_images(5, 1, HW, HW)
    got = big_engine.topiq_score(img)
    with torch.no_grad():
        ref = net(torch.from_numpy(img.astype(np.float32) / 255.0).permute(0, 3, 1, 2)).flatten().numpy()
    assert abs(got[0] - ref[0]) <= 1e-3 * max(abs(ref[0]), 1e-3), (got, ref)


def test_face_detect_fullsize_vs_oracle(big_engine):
    from standins import synthetic_onnx as S
    from facet_amd.face import FaceEngine
    from oracle import face_ref
    det = S.scrfd_like(seed=12, size=640)[0]
    fe = FaceEngine(big_engine, {"det": det}, det_size=(640, 640), max_candidates=4096, max_faces=512)
    imgs = np.random.default_rng(4).integers(0, 256, (3, HW, HW, 3), dtype=np.uint8)
    got = fe.detect(imgs)
    want_det, want_kps = face_ref.scrfd_detect(det, imgs[1], (640, 640))
    g_det, g_kps = got[1]
    assert g_det.shape == want_det.shape and want_det.shape[0] > 0
    assert_int_boxes_match(g_det[:, :4], want_det[:, :4])
    assert np.abs(g_det[:, 4] - want_det[:, 4]).max() < 1e-4 and np.abs(g_kps - want_kps).max() < 5e-2
    faces = fe.get_batch(imgs)                                                         # native path on the same batch
    assert [len(f) for f in faces] == [min(g[0].shape[0], 512) for g in got]
    assert all(np.array_equal(f[0].bbox, g[0][0, :4]) for f, g in zip(faces, got))
    fe.unload()


def test_default_context_handles_fullsize_batches():
    """Engine(0) with no arena size and the default micro-batch (what the mirrors create on their own) must score 1024x1024 batches:
    the default workspace is sized from the free HBM, not a fixed small slab."""
    from facet_amd import Engine
    e = Engine(0)
    e.load_weights(FE_MODEL_TOPIQ, synthetic_state_dict("topiq", seed=3))
    imgs = synthetic_images(9, 9, HW, HW)
    s = e.topiq_score(imgs)
    assert s.shape == (9,) and np.isfinite(s).all()
    e.close()


def test_microbatch_past_4gib_tensors_matches_microbatch_32():
    """Sized for a 288 GB card: with 64 fp32 images of 1024x1024 in flight the 512x512x64 maps are 4.3 GB - past the 32-bit buffer
    addressing of the LDS-DMA kernels. The launcher then issues those layers per image group (conv_split_by_images); every image's
    arithmetic stays what it is at micro-batch 32 up to the tile / split-K choices that depend on the row count (1e-6 on the score)."""
    from facet_amd import Engine
    e = Engine(0, arena_bytes=136 << 30)
    try:
        e.load_weights(FE_MODEL_TOPIQ, synthetic_state_dict("topiq", 5))
        imgs = synthetic_images(21, 66, 1024, 1024)
        d = e.dev_alloc(imgs.nbytes); e.h2d(d, imgs)
        e.set_microbatch(32)
        a = e.topiq_score((d, 66, 1024, 1024))
        e.set_microbatch(64)
        b = e.topiq_score((d, 66, 1024, 1024))
        e.dev_free(d)
        assert np.abs(a - b).max() <= 1e-5 * np.abs(a).max(), float(np.abs(a - b).max())      # throughput of this case: tools/check_large_microbatch.py
    finally:
        e.close()
