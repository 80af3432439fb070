"""GPU parity of U2-Net-P + SAMP-Net against the PINNED oracle: the golden vectors in tests/golden/samp_golden.npz
come from the reference's own classes (models/samp_net.py), see tests/golden/make_samp_golden.py."""
import os

import numpy as np
import pytest
import torch

from facet_amd._lib import FE_MODEL_SAMP, FE_MODEL_U2NETP
from facet_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "samp_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def samp_loaded(engine, gold):
    seed = int(gold["seed_w"])
    engine.load_weights(FE_MODEL_U2NETP, synthetic_state_dict("u2netp", seed))
    engine.load_weights(FE_MODEL_SAMP, synthetic_state_dict("samp_net", seed))
    return seed


def _x(gold, n=2):
    return torch.randn(n, 3, 224, 224, generator=torch.Generator().manual_seed(int(gold["seed_x"]))).numpy()


def test_u2netp_saliency_matches_reference_golden(engine, samp_loaded, gold):
    sal = engine.u2netp_saliency(_x(gold))
    assert sal.shape == (2, 1, 224, 224)
    assert np.abs(sal[:, 0, ::8, ::8] - gold["saliency_ds"]).max() < 1e-4
    assert np.abs(sal[:, 0, 100, :] - gold["saliency_row100"]).max() < 1e-4
    assert np.abs(sal.mean(axis=(1, 2, 3)) - gold["saliency_mean"]).max() < 1e-5


def test_sampnet_outputs_match_reference_golden(engine, samp_loaded, gold):
    engine.set_microbatch(8)
    pw, attrs, dist, sal = engine.samp_forward(_x(gold), want_saliency=True)
    ref_pw = gold["pattern_weights"]
    assert np.abs(pw - ref_pw).max() / np.abs(ref_pw).max() < 1e-3
    assert np.array_equal(pw.argmax(1), ref_pw.argmax(1))              # dominant pattern identical
    assert np.abs(attrs - gold["attributes"]).max() < 1e-3
    assert np.abs(dist - gold["score_dist"]).max() < 1e-3
    raw, raw_ref = (dist * np.arange(1, 6)).sum(1), (gold["score_dist"] * np.arange(1, 6)).sum(1)
    assert np.abs(raw - raw_ref).max() / np.abs(raw_ref).max() < 1e-3   # comp_score before rounding


def test_samp_against_live_oracle_other_seed_and_ragged_batch(engine):
    """Fresh seed, 5 images with micro-batch 2 (2+2+1): HIP vs oracle/sampnet.py run here on CPU."""
    from oracle.sampnet import U2NETP, SAMPNet
    su, ss = synthetic_state_dict("u2netp", 21), synthetic_state_dict("samp_net", 21)
    engine.load_weights(FE_MODEL_U2NETP, su)
    engine.load_weights(FE_MODEL_SAMP, ss)
    u2, sn = U2NETP().eval(), SAMPNet().eval()
    u2.load_state_dict({k: torch.from_numpy(v) for k, v in su.items()})
    sn.load_state_dict({k: torch.from_numpy(v) for k, v in ss.items()})
    x = torch.randn(5, 3, 224, 224, generator=torch.Generator().manual_seed(99))
    with torch.no_grad():
        sal = u2(x)
        rpw, rat, rsd = sn(x, sal)
    engine.set_microbatch(2)
    pw, at, sd, gsal = engine.samp_forward(x.numpy(), want_saliency=True)
    assert np.abs(gsal - sal.numpy()).max() < 1e-4
    assert np.abs(pw - rpw.numpy()).max() / np.abs(rpw.numpy()).max() < 1e-3
    assert np.abs(at - rat.numpy()).max() < 1e-3 and np.abs(sd - rsd.numpy()).max() < 1e-3


def test_micro_batches_of_8_and_40_take_different_kernels_and_agree(engine, samp_loaded, gold):
    """40 images scored as one micro-batch of 40 and as five of 8: SAMP-Net's pattern layers (K = 2592 .. 7524 -> 1024) run on the skinny
    GEMM for <= 32 rows and on the tiled kernel with a K split above; U2-Net-P's 16-channel 3x3 layers run on the halo-tiled 16-column
    kernel (kernels_n16.hip, partial tiles at 56 x 56) once a map has >= 65536 pixels and on the generic kernel below. The results agree
    to fp32 summation-order noise, with the dominant pattern identical."""
    x = torch.randn(40, 3, 224, 224, generator=torch.Generator().manual_seed(77)).numpy()
    engine.set_microbatch(8)
    a = engine.samp_forward(x)
    engine.set_microbatch(64)
    b = engine.samp_forward(x)
    engine.set_microbatch(8)
    for u, v in zip(a, b):
        assert u.shape == v.shape
        assert np.abs(u - v).max() <= 2e-5 * max(1.0, float(np.abs(u).max()))
    assert np.array_equal(a[0].argmax(1), b[0].argmax(1))
