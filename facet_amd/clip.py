"""Drop-in CLIP handle for the reference's `{'model': m, 'preprocess': fn}` (models/model_manager.py:145-148) and
`Facet.get_aesthetic_and_quality_batch` (processing/scorer.py:640-673), backed by libfacet_engine.so.

  fn(PIL) -> Tensor[3,224,224]            open_clip eval transform, on the host with PIL like the reference's loader threads
  m.encode_image(Tensor[B,3,224,224]) -> Tensor[B,768]
  next(m.parameters()).dtype               (callers test for float16, scorer.py:658; multi_pass.py:519)
  m.to()/.cpu()/.half()/.eval()
`m.encode_text(tokens)` (tagger.py:73) runs the text tower on the engine when the checkpoint carries it; the BPE
tokenizer itself ships inside open_clip and is not re-implemented here (pass token ids).
"""
import numpy as np

from ._lib import Engine, FE_MODEL_CLIP, FE_MODEL_AESTHETIC
from .weights import checkpoint_or_synthetic

CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], np.float32)
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711], np.float32)


def clip_preprocess(pil_img):
    """open_clip image_transform(is_train=False): bicubic shorter-side resize to 224, center crop, ToTensor, Normalize."""
    import torch
    from PIL import Image
    img = pil_img.convert('RGB')
    w, h = img.size
    ow, oh = (224, int(224 * h / w)) if w <= h else (int(224 * w / h), 224)
    img = img.resize((ow, oh), Image.BICUBIC)
    top, left = int(round((oh - 224) / 2.0)), int(round((ow - 224) / 2.0))
    a = np.asarray(img.crop((left, top, left + 224, top + 224)), np.float32) / 255.0
    a = (a - CLIP_MEAN) / CLIP_STD
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))


class CLIPImageModel:
    def __init__(self, engine, state_dict):
        self._engine, self._sd = engine, state_dict
        self._resident()

    def _resident(self):
        if not self._engine.loaded(FE_MODEL_CLIP):
            self._engine.load_weights(FE_MODEL_CLIP, self._sd)

    def encode_image(self, x):
        import torch
        self._resident()
        feat = self._engine.clip_encode_image(x.detach().float().cpu().numpy())
        return torch.from_numpy(feat)

    def encode_text(self, tokens):
        """tokens: int tensor/array [n,77] (open_clip tokenizer output) -> Tensor[n,768]; needs a full CLIP checkpoint
        (token_embedding.weight etc.) to have been loaded."""
        import torch
        self._resident()
        tk = tokens.detach().cpu().numpy() if hasattr(tokens, "detach") else np.asarray(tokens)
        return torch.from_numpy(self._engine.clip_encode_text(tk))

    def parameters(self):
        import torch
        yield torch.zeros(1, dtype=torch.float32)  # engine computes in fp32

    def to(self, *a, **k):
        if a and str(a[0]) == 'cpu':
            return self.cpu()
        self._resident()
        return self

    def cpu(self):
        if self._engine.loaded(FE_MODEL_CLIP):
            self._engine.unload(FE_MODEL_CLIP)
        return self

    def half(self):
        return self

    def eval(self):
        return self


def load_clip(engine=None, weights_path=None, synthetic_seed=9, synthetic=False):
    """-> {'model': CLIPImageModel, 'preprocess': fn} like ModelManager._load_clip (model_manager.py:127-148). Without a checkpoint
    path this raises (the reference would download the weights or fail); synthetic=True / FACET_AMD_SYNTHETIC=1 opts into the seeded
    stand-in checkpoint."""
    engine = engine or Engine(0)
    from .pyiqa_scorer import load_checkpoint
    sd = checkpoint_or_synthetic('clip', weights_path, synthetic, synthetic_seed, load_checkpoint)
    return {'model': CLIPImageModel(engine, sd), 'preprocess': clip_preprocess}


class ClipAestheticScorer:
    """`Facet.get_aesthetic_and_quality_batch` (scorer.py:640-673): one engine call for tower + normalise + MLP."""

    def __init__(self, engine, clip_handle, aesthetic_state=None, synthetic_seed=9, synthetic=False, aesthetic_path=None):
        self._engine = engine
        self.model, self.preprocess = clip_handle['model'], clip_handle['preprocess']
        if aesthetic_state is None:     # the reference loads the LAION aesthetic head's weights (scorer.py:560-577) or fails
            from .pyiqa_scorer import load_checkpoint
            aesthetic_state = checkpoint_or_synthetic('aesthetic', aesthetic_path, synthetic, synthetic_seed, load_checkpoint)
        engine.load_weights(FE_MODEL_AESTHETIC, aesthetic_state)

    def get_aesthetic_and_quality_batch(self, pil_images, clip_inputs=None):
        import torch
        inputs = clip_inputs if clip_inputs is not None else torch.stack([self.preprocess(im) for im in pil_images])
        self.model._resident()
        feat, emb, aes = self._engine.clip_encode_image(inputs.detach().float().cpu().numpy(), normalized=True,
                                                        aesthetic=True)
        return [(max(0.0, min(10.0, (float(aes[i]) + 1) * 5)), emb[i].astype(np.float32).tobytes(), None, 'clip-mlp')
                for i in range(len(pil_images))]

    # ---- the single-image entry points of the reference's Facet class (processing/scorer.py:587-638) ------------------------------
    def get_aesthetic_with_embedding(self, image_pil):
        """(aesthetic score, 3072-byte normalised embedding) - scorer.py:603-617."""
        score, emb, _, _ = self.get_aesthetic_and_quality_batch([image_pil])[0]
        return score, emb

    def get_aesthetic_score(self, image_pil):
        """scorer.py:587-601."""
        return self.get_aesthetic_with_embedding(image_pil)[0]

    def get_aesthetic_and_quality(self, pil_img):
        """scorer.py:631-638: (aesthetic, clip_embedding_bytes, None, 'clip-mlp')."""
        return self.get_aesthetic_and_quality_batch([pil_img])[0]

    def score_from_embedding(self, embedding_bytes):
        """Recalculate the aesthetic score from a stored 3072-byte embedding (scorer.py:619-629): the stored, L2-normalised vector goes
        through the same MLP head (the reference does exactly this, although the live path feeds the un-normalised features)."""
        return self.scores_from_embeddings([embedding_bytes])[0]

    def scores_from_embeddings(self, blobs):
        """score_from_embedding for many stored embeddings in one engine call (the reference's recalculation loops call it per row)."""
        vecs = np.stack([np.frombuffer(b, dtype=np.float32) for b in blobs]) if len(blobs) else np.zeros((0, 768), np.float32)
        if len(vecs) == 0:
            return []
        raw = self._engine.aesthetic_score(vecs)
        return [max(0.0, min(10.0, (float(r) + 1) * 5)) for r in raw]
