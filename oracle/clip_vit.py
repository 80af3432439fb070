"""Oracle (TEST INFRASTRUCTURE ONLY): open_clip ViT-L/14 image tower + the aesthetic MLP, torch-CPU fp32.

PARITY UNPINNED for the tower: the reference calls `open_clip.create_model_and_transforms('ViT-L-14',
pretrained='laion2b_s32b_b82k')` (models/model_manager.py:140-143, processing/scorer.py:508-511) and
`model.encode_image` (scorer.py:662); open_clip is a lower-bound-pinned pip dependency (requirements.txt:8,
`open-clip-torch>=2.20.0`), not vendored, not installed here; the reference holds no fixture for its output
(only the 3072-byte blob length, validation/database_validator.py:355-369). This restates the published
VisionTransformer [DEP-KNOWLEDGE]: conv1 (14x14/14, no bias), class_embedding, positional_embedding [257,1024],
ln_pre, 24 residual blocks {ln_1, nn.MultiheadAttention(1024, 16), ln_2, mlp c_fc-GELU(erf)-c_proj}, ln_post on
the class token, `@ proj` [1024,768]. State-dict keys are open_clip's (`visual.*`).
Second opinion: tests/test_oracle_second_opinion.py loads the same weights into HuggingFace transformers' independent CLIP
implementation (installed offline) and gets the same embeddings for both towers.
The aesthetic head IS in-tree: Linear(768,256)-ReLU-Linear(256,1) (processing/scorer.py:579-583), score =
clamp((x+1)*5, 0, 10) (:669) -> pinned by reading that definition.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class _Block(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d)
        self.attn = nn.MultiheadAttention(d, heads)
        self.ln_2 = nn.LayerNorm(d)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(d, d * 4))
        self.mlp.add_module("gelu", nn.GELU())
        self.mlp.add_module("c_proj", nn.Linear(d * 4, d))

    def forward(self, x):  # [L, B, d]
        y = self.ln_1(x)
        x = x + self.attn(y, y, y, need_weights=False)[0]
        return x + self.mlp(self.ln_2(x))


class _Transformer(nn.Module):
    def __init__(self, d, layers, heads):
        super().__init__()
        self.resblocks = nn.ModuleList([_Block(d, heads) for _ in range(layers)])

    def forward(self, x):
        for b in self.resblocks:
            x = b(x)
        return x


class VisionTransformer(nn.Module):
    def __init__(self, width=1024, layers=24, heads=16, patch=14, grid=16, out_dim=768):
        super().__init__()
        self.conv1 = nn.Conv2d(3, width, patch, patch, bias=False)
        self.class_embedding = nn.Parameter(torch.zeros(width))
        self.positional_embedding = nn.Parameter(torch.zeros(grid * grid + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = _Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(torch.zeros(width, out_dim))

    def forward(self, x):
        x = self.conv1(x)
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
        cls = self.class_embedding.to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
        x = torch.cat([cls, x], dim=1) + self.positional_embedding
        x = self.ln_pre(x).permute(1, 0, 2)
        x = self.transformer(x).permute(1, 0, 2)
        return self.ln_post(x[:, 0]) @ self.proj


class TextTransformer(nn.Module):
    """open_clip text tower for ViT-L-14: width 768, 12 layers, 12 heads, context 77, causal mask, features taken at the
    EOT token (argmax of the token ids), `@ text_projection`. Reference call: `encode_text` (models/tagger.py:73)."""

    def __init__(self, width=768, layers=12, heads=12, ctx=77, vocab=49408, out_dim=768):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, width)
        self.positional_embedding = nn.Parameter(torch.zeros(ctx, width))
        self.transformer = _Transformer(width, layers, heads)
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.zeros(width, out_dim))
        self.register_buffer("attn_mask", torch.full((ctx, ctx), float("-inf")).triu_(1), persistent=False)

    def forward(self, text):
        x = self.token_embedding(text) + self.positional_embedding
        x = x.permute(1, 0, 2)
        for blk in self.transformer.resblocks:
            y = blk.ln_1(x)
            x = x + blk.attn(y, y, y, need_weights=False, attn_mask=self.attn_mask)[0]
            x = x + blk.mlp(blk.ln_2(x))
        x = self.ln_final(x.permute(1, 0, 2))
        return x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ self.text_projection


class CLIPImage(nn.Module):
    """Holds the tower under `visual.` like open_clip's CLIP so checkpoint keys match."""

    def __init__(self, **kw):
        super().__init__()
        self.visual = VisionTransformer(**kw)

    def encode_image(self, x):
        return self.visual(x)


def aesthetic_head():
    return nn.Sequential(nn.Linear(768, 256), nn.ReLU(), nn.Linear(256, 1))  # scorer.py:579-583


def aesthetic_score(raw):
    return max(0.0, min(10.0, (float(raw) + 1) * 5))  # scorer.py:669
