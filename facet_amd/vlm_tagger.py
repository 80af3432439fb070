"""Host-side mirror of the reference's VLM tagger (models/vlm_tagger.py) over the engine's Qwen2.5-VL text decoder.

Slice 1 (SURVEY 8(f)-4 / BASELINE configs[4]): the decoder runs in the engine (`fe_vlm_prefill` / `fe_vlm_decode_step`, greedy, bf16 as the
reference loads the model, models/vlm_tagger.py:155-184); what stays here is what the reference also does on the host - the prompt built from
the tag vocabulary (:88-148), the generate loop's bookkeeping and the parsing of the generated text into vocabulary tags (:446-495). The
tokenizer and chat template ship inside the Hugging Face checkpoint (`AutoProcessor.from_pretrained`, :181), which is not available offline
and is not re-implemented: callers pass token ids (and, until the vision tower lands in the next slice, no image rows), exactly as the CLIP
text tower takes token ids (facet_amd/tagger.py).
"""
from typing import Any, Dict, Iterable, List, Optional

import numpy as np

from ._lib import FE_MODEL_VLM

QWEN2_5_VL_7B = dict(n_heads=28, n_kv_heads=4, head_dim=128, rope_theta=1e6, rms_eps=1e-6, mrope_section=(16, 24, 24))


def edit_distance(a: str, b: str) -> int:
    """Levenshtein distance (insert / delete / substitute, unit costs) by a rolling row of the DP table."""
    if not a:
        return len(b)
    if not b:
        return len(a)
    row = np.arange(len(b) + 1)
    for i, ca in enumerate(a, 1):
        diag, row[0] = row[0], i
        for j, cb in enumerate(b, 1):
            diag, row[j] = row[j], min(row[j] + 1, row[j - 1] + 1, diag + (ca != cb))
    return int(row[-1])


class VLMTagger:
    """Same constructor and public surface as the reference class (models/vlm_tagger.py:45-87): `model_config` (model_path,
    vlm_batch_size, max_new_tokens, ...), optional `scoring_config` for the tag vocabulary. `engine` is the facet_amd Engine the decoder
    lives in; `decode` / `encode` are the tokenizer callables of the checkpoint's processor (ids -> text, chat-formatted text -> ids)."""

    def __init__(self, model_config: Dict[str, Any], scoring_config=None, engine=None, decode=None, encode=None):
        self.model_config = model_config
        self.scoring_config = scoring_config
        self.engine = engine
        self.decode, self.encode = decode, encode
        self.model = None
        self.device = "cuda"
        path = model_config.get("model_path", "")
        self.family = "qwen3" if ("Qwen3" in path or "qwen3" in path) else "qwen2_5"
        self.batch_size = model_config.get("vlm_batch_size", 4 if self.family == "qwen3" else 2)
        self.valid_tags = set(scoring_config.get_tag_vocabulary().keys()) if scoring_config else set()
        self._prompt = None

    # -- lifecycle (ModelManager calls load / unload around a pass) ------------------------------------------------------------------
    def load(self, state_dict=None, geometry=None):
        """Commits a Qwen2_5_VLForConditionalGeneration state dict (name -> array) to the engine. geometry: fe_vlm_configure's
        arguments (default Qwen2.5-VL-7B-Instruct)."""
        if self.model is not None:
            return
        if self.family != "qwen2_5":
            raise NotImplementedError("the engine's decoder is Qwen2.5-VL's (Qwen3-VL: not built)")
        if state_dict is None:
            raise FileNotFoundError("no checkpoint: pass the model's state dict (the reference downloads it with from_pretrained, "
                                    "models/vlm_tagger.py:170-176; there is no network here)")
        self.engine.vlm_configure(**(geometry or QWEN2_5_VL_7B))
        self.engine.load_weights(FE_MODEL_VLM, {k: v for k, v in state_dict.items() if not k.startswith("model.visual.")})
        self.model = self.engine

    def unload(self):
        if self.model is not None:
            self.engine.unload(FE_MODEL_VLM)
            self.model = None

    # -- prompt (reference :88-148) -----------------------------------------------------------------------------------------------------
    def _build_prompt(self) -> str:
        if self._prompt is None:
            self._prompt = self._fallback_prompt() if not self.scoring_config else self._vocabulary_prompt()
        return self._prompt

    def _vocabulary_prompt(self) -> str:
        out = ["Analyze this photo and provide semantic tags.", "",
               "Return ONLY a comma-separated list of relevant tags from this exact list:"]
        seen = set()
        for cat in self.scoring_config.get_categories():
            names = [n for n in (cat.get("tags", {}) or {}) if n not in seen]
            if names:
                seen.update(names)
                out.append(f"- {cat['name'].replace('_', ' ').title()}: {', '.join(names)}")
        extra = [n for n in (self.scoring_config.config.get("standalone_tags", {}) or {}) if n not in seen]
        if extra:
            out.append(f"- Other: {', '.join(extra)}")
        out += ["", "Tags:"]
        return "\n".join(out)

    @staticmethod
    def _fallback_prompt() -> str:
        return ("Analyze this photo and provide semantic tags.\n\n"
                "Return ONLY a comma-separated list of relevant tags from these categories:\n"
                "- Scene: landscape, portrait, street, architecture, macro, wildlife, aerial, concert, night, astro, food, sports, travel, "
                "fashion, urban\n"
                "- Subject: person, animal, building, nature, water, sky, mountain, beach, forest, flower, vehicle\n"
                "- Style: black_and_white, silhouette, long_exposure, dramatic, minimalist, vintage, cinematic, abstract\n"
                "- Mood: dramatic, peaceful, energetic, intimate, moody\n\nTags:")

    # -- generation ---------------------------------------------------------------------------------------------------------------------
    def generate_ids(self, input_ids, max_new_tokens: Optional[int] = None, position_ids=None, eos_token_ids: Iterable[int] = ()):
        """Greedy continuation of a batch of equally long prompts: int [n, len] -> int [n, max_new_tokens] (the slice
        `output_ids[:, input_len:]` the reference takes, :262-265 / :363)."""
        if self.model is None:
            raise RuntimeError("VLMTagger.load() first")
        n_new = int(max_new_tokens or self.model_config.get("max_new_tokens", 100))
        return self.engine.vlm_generate(np.asarray(input_ids), n_new, position_ids=position_ids, eos_token_ids=eos_token_ids)

    def tags_from_ids(self, generated_ids, max_tags: int = 5) -> List[List[str]]:
        """Generated ids -> text (the checkpoint's tokenizer) -> vocabulary tags."""
        if self.decode is None:
            raise RuntimeError("no tokenizer: pass decode= (processor.batch_decode of the checkpoint)")
        return [self._parse_tags(self.decode(row), max_tags) for row in np.asarray(generated_ids)]

    # -- parsing (reference :446-495) ---------------------------------------------------------------------------------------------------
    def _parse_tags(self, text: str, max_tags: int) -> List[str]:
        text = text.strip()
        for lead in ("Tags:", "tags:", "Here are the tags:", "The tags are:"):      # applied in this order, each at most once
            if text.startswith(lead):
                text = text[len(lead):].strip()
        found: List[str] = []
        for piece in text.split(","):
            tag = piece.strip().lower().lstrip("0123456789.-) ").strip("\"'")
            if ":" in tag:                                  # a category the model echoed ("art: painting")
                tag = tag.split(":", 1)[1].strip()
            tag = tag.replace(" ", "_")
            if len(tag) <= 1:
                continue
            if self.valid_tags and tag not in self.valid_tags:
                best, best_d = None, 3                      # accept a vocabulary tag within edit distance 2; first best in set order
                for cand in self.valid_tags:
                    d = edit_distance(tag, cand)
                    if d < best_d:
                        best, best_d = cand, d
                if best is not None:
                    tag = best
            if tag not in found:
                found.append(tag)
        return found[:max_tags]
