"""GPU: image-level entry points (uint8 in, scores out) against the oracle fed by the reference's own PIL preprocessing."""
import numpy as np
import pytest
import torch
from PIL import Image

from facet_amd._lib import (FE_MODEL_TOPIQ, FE_MODEL_CLIP, FE_MODEL_AESTHETIC, FE_MODEL_SAMP, FE_MODEL_U2NETP,
                            FE_RECORD_FLOATS)
from facet_amd.weights import synthetic_state_dict, synthetic_images

pytestmark = pytest.mark.gpu
IMNET = (torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))


def _samp_pre(img):  # reference samp_net.py:823-830 (torchvision Resize((224,224)) = PIL bilinear) on an RGB array
    a = np.asarray(Image.fromarray(img).resize((224, 224), Image.BILINEAR), dtype=np.float32) / 255.0
    return (torch.from_numpy(a).permute(2, 0, 1)[None] - IMNET[0]) / IMNET[1]


def _clip_pre(img):  # open_clip eval transform: bicubic shorter side -> 224, center crop, CLIP mean/std
    from oracle.clip_vit import CLIP_MEAN, CLIP_STD
    h, w = img.shape[:2]
    ow, oh = (224, int(224 * h / w)) if w <= h else (int(224 * w / h), 224)
    pil = Image.fromarray(img).resize((ow, oh), Image.BICUBIC)
    top, left = int(round((oh - 224) / 2.0)), int(round((ow - 224) / 2.0))
    a = np.asarray(pil.crop((left, top, left + 224, top + 224)), dtype=np.float32) / 255.0
    m, s = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1), torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    return (torch.from_numpy(a).permute(2, 0, 1)[None] - m) / s


@pytest.fixture(scope="module")
def all_loaded(engine):
    sd = {m: synthetic_state_dict(m, 13) for m in ("topiq", "clip", "aesthetic", "u2netp", "samp_net")}
    engine.load_weights(FE_MODEL_TOPIQ, sd["topiq"])
    engine.load_weights(FE_MODEL_CLIP, sd["clip"])
    engine.load_weights(FE_MODEL_AESTHETIC, sd["aesthetic"])
    engine.load_weights(FE_MODEL_U2NETP, sd["u2netp"])
    engine.load_weights(FE_MODEL_SAMP, sd["samp_net"])
    return sd


def test_ensemble_records_match_oracle(engine, all_loaded):
    from oracle.topiq import CFANet
    from oracle.sampnet import U2NETP, SAMPNet
    from oracle.clip_vit import CLIPImage, aesthetic_head
    sd = all_loaded
    ld = lambda net, d: (net.load_state_dict({k: torch.from_numpy(v) for k, v in d.items()}), net.eval())[1]
    topiq, clip, head = ld(CFANet(), sd["topiq"]), ld(CLIPImage(), sd["clip"]), ld(aesthetic_head(), sd["aesthetic"])
    u2, sn = ld(U2NETP(), sd["u2netp"]), ld(SAMPNet(), sd["samp_net"])
    imgs = synthetic_images(8, 3, 288, 352)  # non-square: exercises the shorter-side resize + center crop
    engine.set_microbatch(2)
    rec, mask = engine.ensemble_score(imgs)
    assert rec.shape == (3, FE_RECORD_FLOATS) and mask == 7
    with torch.no_grad():
        for i in range(3):
            t = float(topiq(torch.from_numpy(imgs[i:i + 1].astype(np.float32) / 255).permute(0, 3, 1, 2)))
            f = clip.encode_image(_clip_pre(imgs[i]))
            a = float(head(f))
            e = torch.nn.functional.normalize(f, dim=-1)[0].numpy()
            xs = _samp_pre(imgs[i])
            pw, at, sdist = sn(xs, u2(xs))
            r = rec[i]
            assert abs(r[0] - t) / max(abs(t), 1e-3) < 1e-3
            assert abs(r[1] - a) / max(abs(a), 1e-3) < 1e-3
            assert np.abs(r[2:10] - pw[0].numpy()).max() / np.abs(pw[0].numpy()).max() < 1e-3
            assert int(np.argmax(r[2:10])) == int(pw[0].argmax())
            assert np.abs(r[10:16] - at[0].numpy()).max() < 1e-3 and np.abs(r[16:21] - sdist[0].numpy()).max() < 1e-3
            assert float((r[21:] * e).sum()) > 1 - 1e-6


def test_samp_images_bgr_flag_and_clip_images(engine, all_loaded):
    imgs = synthetic_images(9, 2, 200, 200)
    pw_rgb, _, _ = engine.samp_score_images(imgs, bgr=False)
    pw_bgr, _, _ = engine.samp_score_images(imgs[..., ::-1].copy(), bgr=True)   # reference converts BGR->RGB first (:916-921)
    assert np.allclose(pw_rgb, pw_bgr, rtol=1e-5, atol=1e-6)
    feat, emb, aes = engine.clip_encode_images(imgs)
    rec, _ = engine.ensemble_score(imgs)
    assert np.allclose(rec[:, 21:], emb, atol=1e-6) and np.allclose(rec[:, 1], aes, rtol=1e-5, atol=1e-6)


def test_crop_gathering_across_microbatches_is_order_preserving():
    """CLIP and SAMP crops are gathered across micro-batches and run in larger chunks (ClipBatcher / SampBatcher in capi.hip): with
    70 images, 16 per micro-batch, the towers run on a full chunk plus a remainder that was shifted inside the gather buffer.
    Every image must get its own result: compare with one-image calls."""
    from facet_amd import Engine
    from facet_amd._lib import FE_MODEL_CLIP, FE_MODEL_AESTHETIC, FE_MODEL_SAMP, FE_MODEL_U2NETP
    from facet_amd.weights import synthetic_state_dict, synthetic_images
    e = Engine(0, arena_bytes=12 << 30)
    e.load_weights(FE_MODEL_CLIP, synthetic_state_dict("clip", 5))
    e.load_weights(FE_MODEL_AESTHETIC, synthetic_state_dict("aesthetic", 5))
    e.load_weights(FE_MODEL_U2NETP, synthetic_state_dict("u2netp", 5))
    e.load_weights(FE_MODEL_SAMP, synthetic_state_dict("samp_net", 5))
    imgs = synthetic_images(11, 70, 96, 80)
    e.set_microbatch(16)
    feat, emb, aes = e.clip_encode_images(imgs)
    pw, at, sd = e.samp_score_images(imgs)
    rec, mask = e.ensemble_score(imgs)
    assert mask == 6
    for i in (0, 15, 16, 47, 62, 63, 64, 69):
        f1, e1, a1 = e.clip_encode_images(imgs[i:i + 1])
        p1, t1, s1 = e.samp_score_images(imgs[i:i + 1])
        assert np.abs(feat[i] - f1[0]).max() <= 2e-5 * np.abs(f1).max() and abs(aes[i] - a1[0]) <= 2e-5 * max(1.0, abs(a1[0]))
        assert np.abs(emb[i] - e1[0]).max() <= 2e-5
        assert np.abs(pw[i] - p1[0]).max() <= 2e-5 * max(1.0, np.abs(p1).max()) and np.abs(sd[i] - s1[0]).max() <= 2e-5
        assert np.abs(rec[i, 21:789] - e1[0]).max() <= 2e-5 and abs(rec[i, 1] - a1[0]) <= 2e-5 * max(1.0, abs(a1[0]))
        assert np.abs(rec[i, 2:10] - p1[0]).max() <= 2e-5 * max(1.0, np.abs(p1).max())
    e.close()
