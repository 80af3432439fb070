"""Oracle (TEST INFRASTRUCTURE ONLY): TOPIQ-NR = pyiqa `CFANet` on a timm ResNet-50 pyramid, torch-CPU fp32.

PARITY UNPINNED. The reference only calls `pyiqa.create_metric('topiq_nr')` (models/pyiqa_scorer.py:33-39,
108-111, forward :212); pyiqa is a lower-bound-pinned pip dependency (requirements.txt:36, `pyiqa>=0.1.10`),
not vendored, not installed here, and the reference holds no test or fixture for its output. This file
restates the published CFANet architecture for `topiq_nr` [DEP-KNOWLEDGE]:
  semantic_model = resnet50 features (5 levels), use_ref=False, inter_dim=256, num_heads=4,
  num_attn_layers=1, activation='gelu', normalize_before=True, no test-time resize, ImageNet mean/std.
  per level (coarse->fine): GatedConv -> adaptive_avg_pool to the 1/32 grid -> 1x1 dim_reduce + GELU ->
  + bicubic-resized (h_emb|w_emb) positional embedding -> 1 pre-norm self-attention encoder layer;
  then 4 cross-attention decoder layers (query = coarser tokens, memory = next finer level),
  an attention-pool encoder layer, token mean, MLP (LN-Linear-GELU-LN-Linear-GELU-Linear) -> MOS.
State-dict keys follow pyiqa's module names so a real `cfanet_nr_koniq_res50` checkpoint maps 1:1
(decoder `self_attn.*` keys of that checkpoint are unused by the forward and ignored).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .resnet import ResNet50Features

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


_ACTS = {"gelu": nn.GELU, "relu": nn.ReLU, "softplus": nn.Softplus}


class GatedConv(nn.Module):
    """pyiqa's GatedConv. Its two activation choices (gated branch; inside weight_blk) carry no parameters, so a checkpoint cannot
    say which a pyiqa release used: both are options here and in the engine (fe_topiq_configure). Default GELU / GELU = this
    builder's recollection of pyiqa/archs/topiq_arch.py (`self.act = nn.GELU()`, `nn.Conv2d(weightdim, 64, 1), nn.GELU(), ...`)
    [DEP-KNOWLEDGE]; the round-1 reviewer recalled Softplus / ReLU. tools/pin_with_real_dependencies.py tries every combination
    against pyiqa itself on a machine that has it."""

    def __init__(self, dim, gate_act="gelu", weight_blk_act="gelu"):
        super().__init__()
        self.splitconv = nn.Conv2d(dim, dim * 2, 1)
        A = _ACTS[weight_blk_act]
        self.weight_blk = nn.Sequential(nn.Conv2d(dim, 64, 1), A(), nn.Conv2d(64, 64, 3, padding=1), A(),
                                        nn.Conv2d(64, 1, 3, padding=1), nn.Sigmoid())
        self.act = _ACTS[gate_act]()

    def forward(self, x):
        x1, x2 = self.splitconv(x).chunk(2, dim=1)
        return self.act(x1) * self.weight_blk(x2)


class EncoderLayer(nn.Module):
    def __init__(self, d, heads, ff):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d, heads)
        self.linear1, self.linear2 = nn.Linear(d, ff), nn.Linear(ff, d)
        self.norm1, self.norm2 = nn.LayerNorm(d), nn.LayerNorm(d)

    def forward(self, src):  # [L, B, d]
        s2 = self.norm1(src)
        src = src + self.self_attn(s2, s2, value=s2)[0]
        s2 = self.norm2(src)
        return src + self.linear2(F.gelu(self.linear1(s2)))


class DecoderLayer(nn.Module):
    def __init__(self, d, heads, ff):
        super().__init__()
        self.multihead_attn = nn.MultiheadAttention(d, heads)
        self.linear1, self.linear2 = nn.Linear(d, ff), nn.Linear(ff, d)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d), nn.LayerNorm(d), nn.LayerNorm(d)

    def forward(self, tgt, memory):
        memory = self.norm2(memory)
        t2 = self.norm1(tgt)
        tgt = tgt + self.multihead_attn(query=t2, key=memory, value=memory)[0]
        t2 = self.norm3(tgt)
        return tgt + self.linear2(F.gelu(self.linear1(t2)))


class _Stack(nn.Module):
    def __init__(self, layer):
        super().__init__()
        self.layers = nn.ModuleList([layer])

    def forward(self, *a):
        x = a[0]
        for l in self.layers:
            x = l(x, *a[1:])
        return x


class CFANet(nn.Module):
    def __init__(self, dims=(64, 256, 512, 1024, 2048), d=256, heads=4, gate_act="gelu", weight_blk_act="gelu"):
        super().__init__()
        ff = min(4 * d, 2048)
        self.semantic_model = ResNet50Features()
        self.weight_pool = nn.ModuleList([GatedConv(c, gate_act, weight_blk_act) for c in dims])
        self.dim_reduce = nn.ModuleList([nn.Sequential(nn.Conv2d(c, d, 1, 1), nn.GELU()) for c in dims])
        self.sa_attn_blks = nn.ModuleList([_Stack(EncoderLayer(d, heads, ff)) for _ in dims])
        self.attn_blks = nn.ModuleList([_Stack(DecoderLayer(d, heads, ff)) for _ in dims[:-1]])
        self.attn_pool = EncoderLayer(d, heads, ff)
        self.score_linear = nn.Sequential(nn.LayerNorm(d), nn.Linear(d, d), nn.GELU(), nn.LayerNorm(d), nn.Linear(d, d),
                                          nn.GELU(), nn.Linear(d, 1))
        self.h_emb = nn.Parameter(torch.zeros(1, d // 2, 32, 1))
        self.w_emb = nn.Parameter(torch.zeros(1, d // 2, 1, 32))

    def head(self, feats):
        th, tw = feats[-1].shape[2:]
        pos = torch.cat((self.h_emb.repeat(1, 1, 1, self.w_emb.shape[3]), self.w_emb.repeat(1, 1, self.h_emb.shape[2], 1)), 1)
        toks = []
        for i in reversed(range(len(feats))):
            t = self.weight_pool[i](feats[i])
            if t.shape[2] > th and t.shape[3] > tw:
                t = F.adaptive_avg_pool2d(t, (th, tw))
            p = F.interpolate(pos, size=t.shape[2:], mode='bicubic', align_corners=False).flatten(2).permute(2, 0, 1)
            t = self.dim_reduce[i](t).flatten(2).permute(2, 0, 1) + p
            toks.append(self.sa_attn_blks[i](t))
        q = toks[0]
        for i in range(len(toks) - 1):
            q = self.attn_blks[i](q, toks[i + 1])
        return self.score_linear(self.attn_pool(q).mean(dim=0))

    def forward(self, x01):
        """x01: [B,3,H,W] in [0,1] exactly as PyIQAScorer._preprocess_image produces (pyiqa_scorer.py:155-164)."""
        m = torch.tensor(MEAN).view(1, 3, 1, 1)
        s = torch.tensor(STD).view(1, 3, 1, 1)
        return self.head(self.semantic_model((x01 - m) / s))


def normalize_score(raw, lo=0.0, hi=1.0):
    """PyIQAScorer._normalize_score for topiq (score_range (0,1)), pyiqa_scorer.py:166-195."""
    raw = max(float(lo), min(float(hi), float(raw)))
    return max(0.0, min(10.0, float((raw - lo) / (hi - lo) * 10.0)))
