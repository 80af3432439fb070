"""GPU: the drop-in Python mirrors (reference call signatures) run on the engine and agree with the raw C-ABI results."""
import numpy as np
import pytest
from PIL import Image

from facet_amd.weights import synthetic_images

pytestmark = pytest.mark.gpu


def test_model_manager_lifecycle_and_scorers(engine):
    from facet_amd.model_manager import ModelManager
    from facet_amd.clip import ClipAestheticScorer
    mm = ModelManager(config=None, engine=engine)
    topiq = mm.load_model_only("topiq")
    samp = mm.load_model_only("samp_net")
    clip = mm.load_model_only("clip")
    assert mm._cache_misses == 3 and set(mm.get_loaded_models()) == {"topiq", "samp_net", "clip"}
    imgs = synthetic_images(3, 3, 160, 192)
    pils = [Image.fromarray(a) for a in imgs]

    s = topiq.score_batch(pils)                      # PyIQAScorer mirror: floats in [0,10]
    assert len(s) == 3 and all(isinstance(v, float) and 0.0 <= v <= 10.0 for v in s)
    raw = engine.topiq_score(imgs)
    assert np.allclose(s, np.clip(raw, 0, 1) * 10, atol=1e-5)
    assert abs(topiq.score_image(pils[1]) - s[1]) < 1e-4
    assert topiq.score_batch([pils[0], "not an image"])[1] == 5.0   # per-image failure -> 5.0 (reference :251-253)

    r = samp.score(imgs[0][..., ::-1].copy())        # ndarray = BGR, like the reference's img_cv
    r2 = samp.score(pils[0])
    assert r["pattern"] == r2["pattern"] and abs(r["comp_score"] - r2["comp_score"]) <= 0.01
    assert set(r) == {"comp_score", "raw_score", "pattern", "pattern_index", "pattern_weights", "score_distribution",
                      "attributes", "power_point_score"}
    assert len(samp.score_batch(pils)) == 3

    sc = ClipAestheticScorer(engine, clip)
    out = sc.get_aesthetic_and_quality_batch(pils)
    assert len(out) == 3 and all(len(o[1]) == 3072 and o[2] is None and o[3] == "clip-mlp" and 0 <= o[0] <= 10 for o in out)
    emb = np.frombuffer(out[0][1], np.float32)
    assert abs(np.linalg.norm(emb) - 1) < 1e-5
    x = clip["preprocess"](pils[0])
    f = clip["model"].encode_image(x[None])
    assert tuple(f.shape) == (1, 768) and str(next(clip["model"].parameters()).dtype) == "torch.float32"
    _, emb_gpu, _ = engine.clip_encode_images(imgs[:1])   # GPU preprocessing path vs host PIL preprocessing path
    assert float((emb_gpu[0] * emb).sum()) > 1 - 1e-5
    # single-image entry points of Facet (scorer.py:587-638) and the recalculation from stored embeddings (:619-629)
    a1, e1 = sc.get_aesthetic_with_embedding(pils[1])
    assert a1 == pytest.approx(out[1][0], abs=1e-4) and e1 == out[1][1] and sc.get_aesthetic_score(pils[1]) == pytest.approx(a1, abs=1e-6)
    assert sc.get_aesthetic_and_quality(pils[2])[1:] == (out[2][1], None, "clip-mlp")
    from facet_amd.weights import synthetic_state_dict
    sd = synthetic_state_dict("aesthetic", 9)                 # the head ClipAestheticScorer loaded: Linear 768-256, ReLU, Linear 256-1
    w1, b1, w2, b2 = (sd[k].astype(np.float64) for k in sorted(sd, key=lambda k: (k.split(".")[0], "bias" in k)))
    for blob, got in zip([o[1] for o in out], sc.scores_from_embeddings([o[1] for o in out])):
        v = np.frombuffer(blob, np.float32).astype(np.float64)
        raw = float((np.maximum(v @ w1.T + b1, 0) @ w2.T + b2)[0])
        assert got == pytest.approx(max(0.0, min(10.0, (raw + 1) * 5)), abs=1e-4)
    assert sc.score_from_embedding(out[0][1]) == sc.scores_from_embeddings([out[0][1]])[0] and sc.scores_from_embeddings([]) == []

    mm.unload_model("topiq")
    assert not engine.loaded(0) and "topiq" not in mm.get_loaded_models()
    again = mm.load_model_only("topiq")              # restored from the host cache (cache hit), same object
    assert again is topiq and mm._cache_hits == 1 and engine.loaded(0)
    assert abs(again.score_image(pils[1]) - s[1]) < 1e-4
    mm.unload_all(); mm.evict_cpu_cache()
    assert mm.get_loaded_models() == []


def test_batched_tag_scoring_matches_per_image_path(engine):
    from facet_amd.tagger import CLIPTagger

    class Cfg:
        def get_tag_vocabulary(self):
            return {f"tag{i}": [f"t{i}a", f"t{i}b"] for i in range(37)}

        def get_art_tags(self):
            return set()
    tg = CLIPTagger(None, "cuda", Cfg())
    rng = np.random.default_rng(3)
    names, _ = tg.prompts()
    tg.set_text_embeddings(names, rng.standard_normal((len(names), 768)))
    embs = rng.standard_normal((9, 768)).astype(np.float32)
    embs /= np.linalg.norm(embs, axis=1, keepdims=True)
    sims = engine.tag_similarities(embs, tg.text_embeddings)
    assert np.abs(sims - embs @ tg.text_embeddings.T).max() < 1e-5
    batched = tg.get_tags_batch([e.tobytes() for e in embs], engine, threshold=0.02, max_tags=5)
    single = [tg.get_tags_from_embedding(e.tobytes(), threshold=0.02, max_tags=5) for e in embs]
    assert batched == single
