"""Drop-in for reference models/tagger.py `CLIPTagger` (tag selection from a stored 768-d embedding).

Same interface (:20-158): CLIPTagger(clip_model, device, config); get_tags_from_embedding(bytes, threshold, max_tags);
get_tags_with_scores; is_artwork; attrs tag_vocabulary, text_embeddings. The text tower is not in the engine yet
(SURVEY §8f-3), so text embeddings come either from `clip_model.encode_text` when the given model has one, or from
`set_text_embeddings()` (precomputed [T,768], L2-normalised). Without them every call returns [] exactly like the
reference does when `clip_model is None` (:38-39, :89-90).
"""
import numpy as np


def bytes_to_embedding(b):
    return np.frombuffer(b, dtype=np.float32)  # reference utils/embedding.py:26


class CLIPTagger:
    def __init__(self, clip_model=None, device='cuda', config=None):
        self.model, self.device, self.config = clip_model, device, config
        self.text_embeddings = None
        self.tag_names = None
        if config:
            self.tag_vocabulary = config.get_tag_vocabulary()
            self.art_tags = config.get_art_tags()
        else:
            self.tag_vocabulary, self.art_tags = {}, set()

    def prompts(self):
        """(tag_names, prompts) flattened like the reference (:60-66): one "a photo of {synonym}" per synonym."""
        names, texts = [], []
        for tag, descs in self.tag_vocabulary.items():
            for d in descs:
                names.append(tag)
                texts.append(f"a photo of {d}")
        return names, texts

    def precompute_text_embeddings(self, tokenizer):
        """Reference _precompute_text_embeddings (:51-75) with an injected tokenizer (open_clip.get_tokenizer('ViT-L-14')
        where available): prompts -> tokens -> clip_model.encode_text on the engine -> L2-normalised rows."""
        if self.model is None:
            return
        names, texts = self.prompts()
        feats = self.model.encode_text(tokenizer(texts))
        feats = feats.detach().cpu().numpy() if hasattr(feats, "detach") else np.asarray(feats)
        self.set_text_embeddings(names, feats)

    def set_text_embeddings(self, tag_names, embeddings):
        e = np.asarray(embeddings, np.float32)
        self.text_embeddings = e / np.linalg.norm(e, axis=-1, keepdims=True)
        self.tag_names = list(tag_names)

    def _tag_scores(self, clip_embedding_bytes):
        sims = self.text_embeddings @ bytes_to_embedding(clip_embedding_bytes).astype(np.float32)
        scores = {}
        for name, s in zip(self.tag_names, sims):
            if name not in scores or s > scores[name]:
                scores[name] = float(s)
        return scores

    def get_tags_from_embedding(self, clip_embedding_bytes, threshold=0.25, max_tags=5):
        if self.text_embeddings is None or clip_embedding_bytes is None:
            return []
        kept = [(t, s) for t, s in self._tag_scores(clip_embedding_bytes).items() if s >= threshold]
        kept.sort(key=lambda x: x[1], reverse=True)
        return [t for t, _ in kept[:max_tags]]

    def get_tags_batch(self, embeddings, engine, threshold=0.25, max_tags=5):
        """Batched form of get_tags_from_embedding: one GPU GEMM for all images ([n,768] array or list of blobs),
        then the reference's per-tag max / threshold / sort / top-k on the host."""
        if self.text_embeddings is None:
            return [[] for _ in embeddings]
        embs = np.stack([bytes_to_embedding(e) if isinstance(e, (bytes, bytearray)) else np.asarray(e, np.float32)
                         for e in embeddings]).astype(np.float32)
        sims = engine.tag_similarities(embs, self.text_embeddings)
        out = []
        for row in sims:
            scores = {}
            for name, s in zip(self.tag_names, row):
                if name not in scores or s > scores[name]:
                    scores[name] = float(s)
            kept = sorted(((t, s) for t, s in scores.items() if s >= threshold), key=lambda x: x[1], reverse=True)
            out.append([t for t, _ in kept[:max_tags]])
        return out

    def get_tags_with_scores(self, clip_embedding_bytes, threshold=0.20):
        if self.text_embeddings is None or clip_embedding_bytes is None:
            return {}
        return {t: round(s, 3) for t, s in self._tag_scores(clip_embedding_bytes).items() if s >= threshold}

    def is_artwork(self, clip_embedding_bytes, threshold=0.24):
        return bool(set(self.get_tags_from_embedding(clip_embedding_bytes, threshold=threshold, max_tags=10)) & self.art_tags)
