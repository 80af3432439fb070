"""GPU: the HIP resampler is BIT-EXACT with PIL.Image.resize (the reference's real preprocessing dependency; PIL is
installed, so this parity is pinned against the library itself, not a restatement)."""
import numpy as np
import pytest
from PIL import Image

pytestmark = pytest.mark.gpu

FILT = {"bilinear": Image.BILINEAR, "bicubic": Image.BICUBIC, "lanczos": Image.LANCZOS}


@pytest.mark.parametrize("filt", ["bilinear", "bicubic", "lanczos"])
@pytest.mark.parametrize("shape,out", [((1024, 1024), (224, 224)), ((600, 2048), (300, 1024)), ((333, 517), (224, 347)),
                                       ((100, 120), (224, 224)), ((224, 500), (224, 224)), ((37, 41), (64, 41))])
def test_resize_bit_exact_with_pil(engine, filt, shape, out):
    rng = np.random.default_rng(hash((filt, shape, out)) % 2 ** 31)
    imgs = rng.integers(0, 256, (2,) + shape + (3,), dtype=np.uint8)
    imgs[1, : shape[0] // 2] = 255  # saturated region: exercises clipping of bicubic/lanczos overshoot
    imgs[1, shape[0] // 2:] = 0
    imgs[1, :, ::7] = rng.integers(0, 256, imgs[1, :, ::7].shape, dtype=np.uint8)
    got = engine.resize_u8(imgs, out[0], out[1], filt)
    for i in range(2):
        ref = np.asarray(Image.fromarray(imgs[i]).resize((out[1], out[0]), FILT[filt]))
        assert np.array_equal(got[i], ref), f"{filt} {shape}->{out}: {np.abs(got[i].astype(int) - ref).max()} max diff"
