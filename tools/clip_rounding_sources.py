"""Which rounding points of an fp16-operand ViT-L/14 carry the error of the aesthetic score? (CPU experiment on the oracle)

The oracle tower (oracle/clip_vit.py, fp32) is re-run with selected tensors rounded to fp16 the way `FE_PRECISION_F16 | RES32` rounds
them: w = weights of the six GEMMs per block, ln = LayerNorm outputs (QKV / fc1 operands), qkv = the projected Q, K, V, p = softmax
probabilities, o = attention output (out_proj operand), h = GELU(fc1) (fc2 operand), x = the residual stream itself (what RES32
avoids). Prints the aesthetic / feature errors per set so the table says where a 2-byte policy loses its 1e-3.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from oracle.clip_vit import CLIPImage, aesthetic_head
from facet_amd.weights import synthetic_state_dict

torch.set_grad_enabled(False)
R = lambda t: t.half().float()


def forward(net, x, rnd, dtype_round=R):
    v = net.visual
    r = (lambda name, t: dtype_round(t) if name in rnd else t)
    x = v.conv1(x)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    cls = v.class_embedding + torch.zeros(x.shape[0], 1, x.shape[-1])
    x = torch.cat([cls, x], dim=1) + v.positional_embedding
    x = r("x", v.ln_pre(x))
    B, L, d = x.shape
    H = 16
    for blk in v.transformer.resblocks:
        y = r("ln", blk.ln_1(x))
        W, bqkv = blk.attn.in_proj_weight, blk.attn.in_proj_bias
        qkv = y @ r("w", W).t() + bqkv
        q, k, vv = qkv.split(d, dim=-1)
        q = r("qkv", q / 8.0); k = r("qkv", k); vv = r("qkv", vv)
        q = q.view(B, L, H, 64).transpose(1, 2); k = k.view(B, L, H, 64).transpose(1, 2); vv = vv.view(B, L, H, 64).transpose(1, 2)
        p = r("p", torch.softmax(q @ k.transpose(-1, -2), dim=-1))
        o = r("o", (p @ vv).transpose(1, 2).reshape(B, L, d))
        x = r("x", x + o @ r("w", blk.attn.out_proj.weight).t() + blk.attn.out_proj.bias)
        y = r("ln", blk.ln_2(x))
        h = r("h", F.gelu(y @ r("w", blk.mlp.c_fc.weight).t() + blk.mlp.c_fc.bias))
        x = r("x", x + h @ r("w", blk.mlp.c_proj.weight).t() + blk.mlp.c_proj.bias)
    pooled = r("ln", v.ln_post(x[:, 0]))
    return pooled @ r("w", v.proj)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    seed = 3
    net = CLIPImage().eval(); net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict("clip", seed).items()})
    head = aesthetic_head().eval(); head.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict("aesthetic", seed).items()})
    x = torch.from_numpy(np.random.default_rng(7).normal(0, 1, (n, 3, 224, 224)).astype(np.float32))
    f0 = forward(net, x, set())
    a0 = (head(f0).flatten() + 1) * 5
    print("aesthetic fp32:", a0.numpy(), " logits? |feat| max", float(f0.abs().max()))
    sets = [("w",), ("ln",), ("qkv",), ("p",), ("o",), ("h",), ("x",), ("w", "ln", "qkv", "p", "o", "h"), ("w", "ln", "qkv", "p", "o", "h", "x"),
            ("ln", "qkv", "p", "o", "h")]
    for s in sets:
        f = forward(net, x, set(s))
        a = (head(f).flatten() + 1) * 5
        rel = ((a - a0).abs() / a0.abs().clamp(min=1.0)).max()
        cos = 1 - (F.normalize(f, dim=-1) * F.normalize(f0, dim=-1)).sum(-1).min()
        print(f"{'+'.join(s):24s} aesthetic_rel {float(rel):.3e}  feat_rel {float((f - f0).abs().max() / f0.abs().max()):.3e}  1-cos {float(cos):.3e}", flush=True)


if __name__ == "__main__":
    main()
