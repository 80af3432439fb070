"""Host-side mirror of the reference's VLM tagger (models/vlm_tagger.py) over the engine's Qwen2.5-VL text decoder.

SURVEY 8(f)-4 / BASELINE configs[4]: the vision tower and the text decoder run in the engine (`fe_vlm_encode_images`, `fe_vlm_prefill_images`,
`fe_vlm_generate`: greedy, bf16 as the reference loads the model, models/vlm_tagger.py:155-184); what stays here is what the reference also
does on the host - the prompt built from the tag vocabulary (:88-148), the index arithmetic transformers derives from `image_grid_thw`
(window order, segment bounds, M-RoPE position ids: `vision_indices`, `rope_index`), the generate loop's bookkeeping and the parsing of the
generated text into vocabulary tags (:446-495). The tokenizer, chat template and image processor (resize to a multiple of 28, normalise,
patchify) ship inside the Hugging Face checkpoint (`AutoProcessor.from_pretrained`, :181), which is not available offline and is not
re-implemented: callers pass the processor's tensors - `input_ids`, `pixel_values`, `image_grid_thw` - exactly as the CLIP text tower takes
token ids (facet_amd/tagger.py).
"""
from typing import Any, Dict, Iterable, List, Optional

import numpy as np

from ._lib import FE_MODEL_VLM

QWEN2_5_VL_7B = dict(n_heads=28, n_kv_heads=4, head_dim=128, rope_theta=1e6, rms_eps=1e-6, mrope_section=(16, 24, 24))


def vision_indices(grid_thw, spatial_merge_size: int = 2, window_size: int = 112, patch_size: int = 14):
    """The index arrays of the vision tower for images of `grid_thw` [n_images, 3] (t, h, w in patches): numpy restatement of
    transformers.vision_utils.get_vision_position_ids / get_vision_window_index / get_vision_cu_seqlens (what
    Qwen2_5_VisionTransformerPretrainedModel.forward derives from image_grid_thw). Returns a dict for Engine.vlm_encode_images:
    patch_pos_hw [n, 2] (row, column of every patch, in WINDOW order), window_index [n / m^2] (raster merge-block index at each
    window-order slot), cu_window_seqlens, cu_seqlens (segment bounds in patches)."""
    grid = np.asarray(grid_thw, dtype=np.int64).reshape(-1, 3)
    m, unit = spatial_merge_size, spatial_merge_size ** 2
    win = window_size // spatial_merge_size // patch_size          # window side in merge blocks (4)
    pos, widx, cu_win, cu_full, base = [], [], [0], [0], 0
    for t, h, w in grid:
        hh, ww = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        blk = lambda a: a.reshape(h // m, m, w // m, m).transpose(0, 2, 1, 3).reshape(-1)      # block-major over the m x m merge blocks
        pos.append(np.tile(np.stack([blk(hh), blk(ww)], -1), (t, 1)))
        gh, gw = h // m, w // m
        idx = np.arange(t * gh * gw).reshape(t, gh, gw)
        ph, pw = win - gh % win, win - gw % win                    # (a full extra window of padding when the side divides: as transformers)
        nh, nw = (gh + ph) // win, (gw + pw) // win
        padded = np.pad(idx, ((0, 0), (0, ph), (0, pw)), constant_values=-100)
        padded = padded.reshape(t, nh, win, nw, win).transpose(0, 1, 3, 2, 4).reshape(t, nh * nw, win, win)
        seqlens = (padded != -100).sum((2, 3)).reshape(-1)
        flat = padded.reshape(-1)
        widx.append(flat[flat != -100] + base)
        cu_win.extend((np.cumsum(seqlens) * unit + cu_win[-1]).tolist())
        base += t * gh * gw
        for _ in range(t):                                         # full attention: one segment per frame
            cu_full.append(cu_full[-1] + h * w)
    window_index = np.concatenate(widx).astype(np.int32)
    cu_win = np.asarray(cu_win, np.int32)
    cu_win = cu_win[np.concatenate([[True], np.diff(cu_win) != 0])]          # unique_consecutive: drops the empty padded windows
    pos = np.concatenate(pos, 0)                                             # raster (block-major) order
    n = pos.shape[0]
    pos_w = pos.reshape(n // unit, unit, 2)[window_index].reshape(n, 2)      # rows regrouped window by window, like the hidden states
    return {"patch_pos_hw": pos_w.astype(np.int32), "window_index": window_index, "cu_window_seqlens": cu_win,
            "cu_seqlens": np.asarray(cu_full, np.int32)}


def rope_index(input_ids, grid_thw, image_token_id: int, spatial_merge_size: int = 2):
    """M-RoPE position ids [3, n_seq, len] of prompts with image placeholders: numpy restatement of Qwen2_5_VLModel.get_rope_index for
    still images and unpadded sequences (what the reference's processor + generate compute): text tokens count up on all three axes; a
    run of <|image_pad|> tokens takes (start, start + row, start + column) over its merged grid, and the next text token continues at
    start + max(rows, columns). Returns (position_ids, next_position [n_seq] = the position of the first generated token)."""
    ids = np.asarray(input_ids)
    grids = iter(np.asarray(grid_thw, dtype=np.int64).reshape(-1, 3))
    out = np.zeros((3,) + ids.shape, np.int32)
    nxt = np.zeros(ids.shape[0], np.int32)
    for b, row in enumerate(ids):
        cur, i, cols = 0, 0, []
        while i < len(row):
            if row[i] == image_token_id:
                t, h, w = next(grids)
                gh, gw = int(h) // spatial_merge_size, int(w) // spatial_merge_size
                n = int(t) * gh * gw
                if not (row[i:i + n] == image_token_id).all():
                    raise ValueError("a run of image placeholder tokens does not match its grid")
                hh, ww = np.meshgrid(np.arange(gh), np.arange(gw), indexing="ij")
                tt = np.repeat(np.arange(int(t)), gh * gw)
                cols.append(np.stack([tt + cur, np.tile(hh.reshape(-1), int(t)) + cur, np.tile(ww.reshape(-1), int(t)) + cur]))
                cur += max(gh, gw)
                i += n
            else:
                cols.append(np.full((3, 1), cur))
                cur += 1
                i += 1
        p = np.concatenate(cols, 1)
        out[:, b] = p
        nxt[b] = p.max() + 1
    return out, nxt


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
        geometry = dict(geometry or QWEN2_5_VL_7B)
        vis = {k: geometry.pop(k) for k in ("vis_heads", "fullatt_block_indexes") if k in geometry}
        self.engine.vlm_configure(**geometry)
        self.engine.vlm_vision_configure(vis.get("vis_heads", 16), vis.get("fullatt_block_indexes", (7, 15, 23, 31)))
        self.engine.load_weights(FE_MODEL_VLM, state_dict)      # model.language_model.*, lm_head.weight and (when present) model.visual.*
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
    def generate_with_images(self, input_ids, pixel_values, image_grid_thw, image_token_id: int, max_new_tokens: Optional[int] = None,
                             eos_token_ids: Iterable[int] = ()):
        """`self.model.generate(**processor(text=..., images=...), max_new_tokens=..., do_sample=False)` (reference :245-259, :346-360) on the
        processor's tensors: input_ids int [n, len] with the <|image_pad|> runs in place, pixel_values [n_patches, 1176], image_grid_thw
        [n_images, 3]. The vision tower encodes all images of the batch in one call, their embeddings replace the placeholder rows, the
        decoder prefills with the M-RoPE positions of get_rope_index and decodes greedily. -> int [n, max_new_tokens]."""
        if self.model is None:
            raise RuntimeError("VLMTagger.load() first")
        ids = np.asarray(input_ids)
        idx = vision_indices(image_grid_thw)
        self.engine.vlm_encode_images(pixel_values, idx["patch_pos_hw"], idx["window_index"], idx["cu_window_seqlens"], idx["cu_seqlens"], want_embeds=False)
        pos, _ = rope_index(ids, image_grid_thw, image_token_id)
        rows = np.flatnonzero(ids.reshape(-1) == image_token_id).astype(np.int32)
        n_new = int(max_new_tokens or self.model_config.get("max_new_tokens", 100))
        return self.engine.vlm_generate(ids, n_new, position_ids=pos, eos_token_ids=eos_token_ids, image_rows=rows)

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
