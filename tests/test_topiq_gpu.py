"""GPU parity of the TOPIQ path (ResNet-50 pyramid + CFANet head) against the torch-CPU oracle.

Oracle status: "parity unpinned" (architecture restated from pyiqa/timm public definitions, no reference
fixtures exist — see oracle/__init__.py). What IS checked: HIP result == oracle result on the same seeded
weights and inputs, within 1e-3 relative (BASELINE.json north_star tolerance).
"""
import numpy as np
import pytest
import torch

from facet_amd._lib import FE_MODEL_TOPIQ
from facet_amd.weights import synthetic_state_dict, synthetic_images

pytestmark = pytest.mark.gpu

MEAN = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)


@pytest.fixture(scope="module")
def topiq_loaded(engine):
    sd = synthetic_state_dict("topiq", seed=3)
    engine.load_weights(FE_MODEL_TOPIQ, sd)
    return sd


def _oracle_feats(sd, imgs):
    from oracle.resnet import ResNet50Features
    net = ResNet50Features().eval()
    bb = {k[len("semantic_model."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("semantic_model.")}
    net.load_state_dict(bb, strict=True)
    x = torch.from_numpy(imgs.astype(np.float32) / 255.0).permute(0, 3, 1, 2)  # pyiqa_scorer.py:155-158
    with torch.no_grad():
        return net((x - MEAN) / STD)


@pytest.mark.parametrize("hw", [(128, 160), (256, 256)])
def test_resnet50_pyramid(engine, topiq_loaded, hw):
    imgs = synthetic_images(1, 3, *hw)
    ref = _oracle_feats(topiq_loaded, imgs)
    engine.set_microbatch(2)  # exercises the ragged last micro-batch (3 = 2 + 1)
    for level in range(5):
        got = engine.topiq_features(imgs, level)
        r = ref[level].numpy()
        assert got.shape == r.shape
        err = np.abs(got - r).max() / np.abs(r).max()
        assert err < 1e-3, f"level {level}: {err:.3e}"


def _oracle_scores(sd, imgs):
    from oracle.topiq import CFANet
    net = CFANet().eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    x = torch.from_numpy(imgs.astype(np.float32) / 255.0).permute(0, 3, 1, 2)
    with torch.no_grad():
        return net(x).flatten().numpy()


@pytest.mark.parametrize("hw,n", [((128, 160), 3), ((224, 224), 2), ((256, 320), 2)])
def test_topiq_score_matches_oracle(engine, topiq_loaded, hw, n):
    """Raw MOS within 1e-3 relative of the oracle (north_star tolerance), ragged micro-batches, odd token counts
    (224 -> 7x7 = 49 tokens exercises the padded-key attention path)."""
    imgs = synthetic_images(5, n, *hw)
    ref = _oracle_scores(topiq_loaded, imgs)
    engine.set_microbatch(2)
    got = engine.topiq_score(imgs)
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)
    assert rel.max() < 1e-3, f"got {got} ref {ref} rel {rel}"


def test_topiq_device_resident_input_and_batch_invariance(engine, topiq_loaded):
    """Same scores whether images arrive from host or are resident in HBM, and independent of micro-batching."""
    imgs = synthetic_images(6, 5, 128, 128)
    engine.set_microbatch(5)
    a = engine.topiq_score(imgs)
    d = engine.dev_alloc(imgs.nbytes)
    engine.h2d(d, imgs)
    engine.set_microbatch(2)
    b = engine.topiq_score((d, 5, 128, 128))
    engine.dev_free(d)
    assert np.allclose(a, b, rtol=1e-5, atol=1e-6)
