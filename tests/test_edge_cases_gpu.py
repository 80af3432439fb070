"""GPU: error behaviour and edge cases of the C-ABI entry points (the reference's failure conventions)."""
import ctypes as C

import numpy as np
import pytest
import torch

from facet_amd import Engine, EngineError
from facet_amd._lib import FE_MODEL_TOPIQ, FE_MODEL_CLIP, FE_MODEL_SAMP
from facet_amd.weights import synthetic_state_dict, synthetic_images

pytestmark = pytest.mark.gpu


def test_not_loaded_and_bad_arguments_return_errors_not_crashes(engine):
    engine.unload(FE_MODEL_TOPIQ)
    with pytest.raises(EngineError, match="not loaded"):
        engine.topiq_score(synthetic_images(1, 1, 64, 64))
    engine.unload(FE_MODEL_CLIP)
    with pytest.raises(EngineError, match="not loaded"):
        engine.clip_encode_image(np.zeros((1, 3, 224, 224), np.float32))
    with pytest.raises(EngineError):
        engine.load_weights(FE_MODEL_TOPIQ, {"semantic_model.conv1.weight": np.zeros((64, 3, 7, 7), np.float32)})  # missing tensors
    assert "missing weight tensor" in engine.lib.fe_last_error(engine.h).decode()
    engine.load_weights(FE_MODEL_TOPIQ, synthetic_state_dict("topiq", 3))
    with pytest.raises(EngineError, match="bad arguments"):
        engine.topiq_score(np.zeros((1, 16, 16, 3), np.uint8))       # smaller than the 32-px pyramid stride
    with pytest.raises((EngineError, ValueError)):
        engine.conv2d(np.zeros((1, 8, 4, 4), np.float32), np.zeros((8, 8, 7, 7), np.float32))  # empty output


def test_arena_exhaustion_is_a_clean_error_and_recoverable():
    small = Engine(0, arena_bytes=64 << 20)
    small.load_weights(FE_MODEL_TOPIQ, synthetic_state_dict("topiq", 3))
    with pytest.raises(EngineError, match="arena exhausted"):
        small.topiq_score(synthetic_images(1, 2, 512, 512))
    got = small.topiq_score(synthetic_images(1, 1, 64, 96))           # still usable afterwards
    assert np.isfinite(got).all()
    small.close()


@pytest.mark.parametrize("hw", [(97, 131), (250, 333), (64, 64), (33, 500)])
def test_topiq_arbitrary_sizes_match_oracle(engine, hw):
    """Sizes that are not multiples of 32: ceil-mode pyramid, adaptive pooling with ragged windows, odd token grids."""
    from oracle.topiq import CFANet
    sd = synthetic_state_dict("topiq", 3)
    engine.load_weights(FE_MODEL_TOPIQ, sd)
    net = CFANet().eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    imgs = synthetic_images(21, 2, *hw)
    with torch.no_grad():
        ref = net(torch.from_numpy(imgs.astype(np.float32) / 255).permute(0, 3, 1, 2)).flatten().numpy()
    engine.set_microbatch(8)
    got = engine.topiq_score(imgs)
    assert (np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)).max() < 1e-3


def test_single_image_and_microbatch_larger_than_batch(engine):
    engine.load_weights(FE_MODEL_TOPIQ, synthetic_state_dict("topiq", 3))
    imgs = synthetic_images(22, 3, 96, 96)
    engine.set_microbatch(64)
    a = engine.topiq_score(imgs)
    engine.set_microbatch(1)
    b = engine.topiq_score(imgs)
    one = engine.topiq_score(imgs[1:2])
    assert np.allclose(a, b, rtol=1e-5, atol=1e-6) and np.allclose(one, a[1:2], rtol=1e-5, atol=1e-6)
    with pytest.raises(EngineError):
        engine.set_microbatch(0)


def test_reload_cycles_do_not_leak_device_memory(engine):
    sd = synthetic_state_dict("samp_net", 5)
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(4):
        engine.load_weights(FE_MODEL_SAMP, sd)
        engine.unload(FE_MODEL_SAMP)
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (64 << 20), f"leaked {(free0 - free1) >> 20} MiB over 4 load/unload cycles"


def test_topiq_long_edge_cap_matches_reference_preprocessing(engine):
    """Images with a long edge > 1024 are LANCZOS-reduced on the GPU exactly like PyIQAScorer._preprocess_image does with
    PIL on the host (reference pyiqa_scorer.py:131-153): engine(raw big image) == engine(PIL-resized image)."""
    from PIL import Image
    engine.load_weights(FE_MODEL_TOPIQ, synthetic_state_dict("topiq", 3))
    big = synthetic_images(31, 2, 700, 1400)
    scale = 1024 / 1400
    small = np.stack([np.asarray(Image.fromarray(a).resize((int(1400 * scale), int(700 * scale)), Image.LANCZOS)) for a in big])
    engine.set_microbatch(2)
    a = engine.topiq_score(big)
    b = engine.topiq_score(small)
    assert np.array_equal(a, b)
