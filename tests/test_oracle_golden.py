"""CPU: pin the oracle. oracle/sampnet.py (U2NETP + SAMPNet restatement) must reproduce the golden vectors that
tests/golden/make_samp_golden.py captured from the REFERENCE's own classes (models/samp_net.py), and the
post-processing helpers must reproduce the reference's arithmetic on hand-made inputs."""
import os

import numpy as np
import pytest
import torch

from facet_amd.weights import synthetic_state_dict, SPECS

GOLD = os.path.join(os.path.dirname(__file__), "golden", "samp_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def samp_outputs(gold):
    from oracle.sampnet import U2NETP, SAMPNet
    seed = int(gold["seed_w"])
    u2, sn = U2NETP().eval(), SAMPNet().eval()
    u2.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict("u2netp", seed).items()}, strict=True)
    sn.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict("samp_net", seed).items()}, strict=True)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(int(gold["seed_x"])))
    with torch.no_grad():
        sal = u2(x)
        pw, attrs, dist = sn(x, sal)
    return sal, pw, attrs, dist


def test_u2netp_matches_reference_golden(gold, samp_outputs):
    sal = samp_outputs[0]
    assert np.abs(sal[:, 0, ::8, ::8].numpy() - gold["saliency_ds"]).max() < 1e-6
    assert np.abs(sal[:, 0, 100, :].numpy() - gold["saliency_row100"]).max() < 1e-6
    assert np.abs(sal.mean(dim=(1, 2, 3)).numpy() - gold["saliency_mean"]).max() < 1e-6


def test_sampnet_matches_reference_golden(gold, samp_outputs):
    _, pw, attrs, dist = samp_outputs
    assert np.abs(pw.numpy() - gold["pattern_weights"]).max() < 1e-5
    assert np.abs(attrs.numpy() - gold["attributes"]).max() < 1e-6
    assert np.abs(dist.numpy() - gold["score_dist"]).max() < 1e-6
    assert np.allclose(dist.sum(1).numpy(), 1.0, atol=1e-5)


def test_samp_postprocess_contract(gold):
    """Dict keys / rounding of SAMPNetScorer.score (reference samp_net.py:957-989)."""
    from oracle.sampnet import samp_postprocess, COMPOSITION_PATTERNS
    r = samp_postprocess(gold["pattern_weights"][0], gold["attributes"][0], gold["score_dist"][0])
    assert set(r) == {"comp_score", "raw_score", "pattern", "pattern_index", "pattern_weights", "score_distribution",
                      "attributes", "power_point_score"}
    assert r["pattern"] == COMPOSITION_PATTERNS[int(np.argmax(gold["pattern_weights"][0]))]
    raw = float(np.sum(np.arange(1, 6) * gold["score_dist"][0]))
    assert r["raw_score"] == round(raw, 2) and r["comp_score"] == round((raw - 1) / 4 * 10, 2)
    assert abs(sum(r["pattern_weights"].values()) - 1.0) < 1e-5 and len(r["attributes"]) == 6


def test_topiq_normalize_score_matches_reference_formula():
    """PyIQAScorer._normalize_score for topiq: clamp to [0,1], x10 (reference pyiqa_scorer.py:166-195)."""
    from oracle.topiq import normalize_score
    assert normalize_score(0.37) == pytest.approx(3.7)
    assert normalize_score(-0.2) == 0.0 and normalize_score(1.7) == 10.0


@pytest.mark.parametrize("model,cls", [("topiq", "oracle.topiq:CFANet"), ("u2netp", "oracle.sampnet:U2NETP"),
                                       ("samp_net", "oracle.sampnet:SAMPNet")])
def test_synthetic_checkpoint_keys_match_oracle_modules(model, cls):
    """The product-side key/shape specs (facet_amd.weights) and the oracle modules agree exactly."""
    import importlib
    mod, name = cls.split(":")
    net = getattr(importlib.import_module(mod), name)()
    want = {k: tuple(v.shape) for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}
    have = {n: tuple(s) for n, s, _ in SPECS[model]()}
    assert want == have


def test_synthetic_checkpoint_is_deterministic():
    a = synthetic_state_dict("aesthetic", 5)
    b = synthetic_state_dict("aesthetic", 5)
    c = synthetic_state_dict("aesthetic", 6)
    assert all(np.array_equal(a[k], b[k]) for k in a) and not np.array_equal(a["0.weight"], c["0.weight"])
