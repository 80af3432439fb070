"""GPU parity of the CLIP ViT-L/14 image tower + aesthetic head against the torch-CPU oracle ("parity unpinned":
oracle/clip_vit.py restates open_clip's published architecture; see its header)."""
import numpy as np
import pytest
import torch

from facet_amd._lib import FE_MODEL_CLIP, FE_MODEL_AESTHETIC
from facet_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def clip_loaded(engine):
    sd = synthetic_state_dict("clip", seed=9)
    sa = synthetic_state_dict("aesthetic", seed=9)
    engine.load_weights(FE_MODEL_CLIP, sd)
    engine.load_weights(FE_MODEL_AESTHETIC, sa)
    return sd, sa


def test_clip_features_embedding_and_aesthetic(engine, clip_loaded):
    from oracle.clip_vit import CLIPImage, aesthetic_head
    sd, sa = clip_loaded
    net = CLIPImage().eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    head = aesthetic_head().eval()
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sa.items()}, strict=True)
    x = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        ref = net.encode_image(x)
        ref_n = torch.nn.functional.normalize(ref, dim=-1)
        ref_a = head(ref).flatten()
    engine.set_microbatch(2)  # 2 + 1
    feat, emb, aes = engine.clip_encode_image(x.numpy(), normalized=True, aesthetic=True)
    assert feat.shape == (3, 768)
    assert np.abs(feat - ref.numpy()).max() / np.abs(ref.numpy()).max() < 1e-3
    cos = (emb * ref_n.numpy()).sum(1)
    assert cos.min() > 1 - 1e-6 and np.abs(np.linalg.norm(emb, axis=1) - 1).max() < 1e-5
    assert emb.astype(np.float32).tobytes().__len__() == 3 * 3072   # 3072-byte blob per image (validator :355-369)
    assert np.abs(aes - ref_a.numpy()).max() / max(np.abs(ref_a.numpy()).max(), 1e-3) < 1e-3


def test_clip_text_tower_matches_oracle(engine):
    """Causal text transformer + EOT pooling + projection vs oracle/clip_vit.py TextTransformer (parity unpinned)."""
    from oracle.clip_vit import TextTransformer
    sd = synthetic_state_dict("clip", seed=9)
    st = synthetic_state_dict("clip_text", seed=9)
    engine.load_weights(FE_MODEL_CLIP, {**sd, **st})
    net = TextTransformer().eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
    rng = np.random.default_rng(5)
    tokens = np.zeros((5, 77), np.int64)
    for i, L in enumerate([3, 9, 20, 76, 40]):           # SOT, words..., EOT (largest id), zero padding
        tokens[i, 0] = 49406
        tokens[i, 1:L] = rng.integers(1, 40000, L - 1)
        tokens[i, L] = 49407
    with torch.no_grad():
        ref = net(torch.from_numpy(tokens)).numpy()
    got = engine.clip_encode_text(tokens)
    assert got.shape == (5, 768)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-3
    # image tower still served from the same checkpoint
    assert engine.clip_encode_image(np.zeros((1, 3, 224, 224), np.float32)).shape == (1, 768)
