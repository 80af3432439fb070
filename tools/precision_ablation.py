"""Error of the FINAL scores per model and precision policy (SURVEY 8(d) parity gate: 1e-3 on final scores).

  python tools/precision_ablation.py [--out gpurun_out/precision_ablation.json] [--oracle 2]

Policies: {bf16, f16} x {2-byte residual stream, fp32 residual stream ('+r32')}, per model (TOPIQ, CLIP ViT-L/14 + aesthetic MLP,
U2-Net-P + SAMP-Net). The fp32 engine is the dense reference (it is held to the CPU oracle at 1e-3 / 1e-5 by the test-suite);
`--oracle N` additionally runs the torch-CPU oracle on the first N inputs of every model so the table also carries engine-vs-oracle
columns for the policy candidates.
Final scores (what the reference stores): TOPIQ MOS x 10 (models/pyiqa_scorer.py:166-195), aesthetic (x + 1) * 5
(processing/scorer.py:669), comp_score = (sum k p_k - 1) / 4 * 10 and the argmax pattern (models/samp_net.py:957-989), the
L2-normalised CLIP embedding (cosine). Errors are relative to max(|ref|, floor) with the floors of the parity tests.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FACET_AMD_SYNTHETIC"] = "1"
import numpy as np

from facet_amd import Engine
from facet_amd._lib import FE_MODEL_TOPIQ, FE_MODEL_CLIP, FE_MODEL_AESTHETIC, FE_MODEL_SAMP, FE_MODEL_U2NETP
from facet_amd.weights import synthetic_state_dict, synthetic_images

IDS = {"topiq": FE_MODEL_TOPIQ, "clip": FE_MODEL_CLIP, "aesthetic": FE_MODEL_AESTHETIC, "u2netp": FE_MODEL_U2NETP, "samp_net": FE_MODEL_SAMP}
POLICIES = ["f32", "bf16", "f16", "bf16+r32", "f16+r32"]


def comp_score(dist):
    raw = (dist * np.arange(1, 6, dtype=np.float64)).sum(1)
    return np.clip((raw - 1.0) / 4.0 * 10.0, 0.0, 10.0)


def run_policy(policy, seed, topiq_sets, clip_x, samp_x, timing):
    e = Engine(0, arena_bytes=40 << 30, precision=policy)
    out = {}
    try:
        for n in ("topiq", "clip", "aesthetic", "u2netp", "samp_net"):
            e.load_weights(IDS[n], synthetic_state_dict(n, seed))
        e.set_microbatch(4)
        t = []
        for imgs in topiq_sets:
            t.append(e.topiq_score(imgs))
        out["topiq_raw"] = np.concatenate(t)
        feat, emb, aes = e.clip_encode_image(clip_x, normalized=True, aesthetic=True)
        out["clip_emb"], out["clip_feat"], out["aes_raw"] = emb, feat, aes
        pw, at, dist, sal = e.samp_forward(samp_x, want_saliency=True)
        out["pw"], out["attr"], out["dist"], out["sal"] = pw, at, dist, sal
        if timing:      # a rough speed column (one warm pass each; the real numbers are bench.py's)
            big = topiq_sets[-1]
            e.set_microbatch(8)
            e.topiq_score(big)
            e.timer_start(); e.topiq_score(big); out["t_topiq_ms_per_img"] = e.timer_stop() / len(big)
            e.clip_encode_image(clip_x, normalized=True, aesthetic=True)
            e.timer_start(); e.clip_encode_image(clip_x, normalized=True, aesthetic=True); out["t_clip_ms_per_img"] = e.timer_stop() / len(clip_x)
            e.timer_start(); e.samp_forward(samp_x); out["t_samp_ms_per_img"] = e.timer_stop() / len(samp_x)
    finally:
        e.close()
    return out


def errors(got, ref):
    """Final-score errors of one policy against a reference dict (same keys; ref may cover only the first rows)."""
    n = lambda k: min(len(got[k]), len(ref[k]))
    r = {}
    k = n("topiq_raw")
    # the stored score is clamp(raw, 0, 1) * 10; the error is taken on raw (the clamp would hide it), relative with the tests' floor
    r["topiq_rel"] = float((np.abs(got["topiq_raw"][:k] - ref["topiq_raw"][:k]) / np.maximum(np.abs(ref["topiq_raw"][:k]), 1e-3)).max())
    k = n("aes_raw")
    ga, ra = (got["aes_raw"][:k] + 1) * 5, (ref["aes_raw"][:k] + 1) * 5
    r["aesthetic_rel"] = float((np.abs(ga - ra) / np.maximum(np.abs(ra), 1.0)).max())
    r["clip_one_minus_cos"] = float((1.0 - (got["clip_emb"][:k].astype(np.float64) * ref["clip_emb"][:k]).sum(1)).max())
    r["clip_emb_maxabs"] = float(np.abs(got["clip_emb"][:k] - ref["clip_emb"][:k]).max())
    k = n("dist")
    gc, rc = comp_score(got["dist"][:k]), comp_score(ref["dist"][:k])
    r["comp_score_rel"] = float((np.abs(gc - rc) / np.maximum(np.abs(rc), 1.0)).max())
    r["comp_score_abs"] = float(np.abs(gc - rc).max())
    r["pattern_argmax_same"] = bool((got["pw"][:k].argmax(1) == ref["pw"][:k].argmax(1)).all())
    r["pattern_weights_maxabs"] = float(np.abs(got["pw"][:k] - ref["pw"][:k]).max())
    r["attributes_maxabs"] = float(np.abs(got["attr"][:k] - ref["attr"][:k]).max())
    r["score_dist_maxabs"] = float(np.abs(got["dist"][:k] - ref["dist"][:k]).max())
    r["saliency_maxabs"] = float(np.abs(got["sal"][:k] - ref["sal"][:k].reshape(got["sal"][:k].shape)).max())
    return r


def oracle_outputs(seed, topiq_sets, clip_x, samp_x, n):
    import torch
    import torch.nn.functional as F
    from oracle.topiq import CFANet
    from oracle.clip_vit import CLIPImage, aesthetic_head
    from oracle.sampnet import U2NETP, SAMPNet
    ld = lambda net, name: (net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(name, seed).items()}), net.eval())[1]
    out = {}
    with torch.no_grad():
        net = ld(CFANet(), "topiq")
        t = []
        for imgs in topiq_sets:      # the first n of every size
            for a in imgs[:n]:
                t.append(net(torch.from_numpy(a[None].astype(np.float32) / 255.0).permute(0, 3, 1, 2)).flatten().numpy())
        out["topiq_sets"] = [np.concatenate(t[i * n:(i + 1) * n]) for i in range(len(topiq_sets))]
        clip, head = ld(CLIPImage(), "clip"), ld(aesthetic_head(), "aesthetic")
        f = clip.encode_image(torch.from_numpy(clip_x[:n]))
        out["clip_emb"], out["clip_feat"], out["aes_raw"] = F.normalize(f, dim=-1).numpy(), f.numpy(), head(f).flatten().numpy()
        u2, sn = ld(U2NETP(), "u2netp"), ld(SAMPNet(), "samp_net")
        xs = torch.from_numpy(samp_x[:n])
        s = u2(xs)
        p, a, d = sn(xs, s)
        s = s[0] if isinstance(s, (tuple, list)) else s
        out["pw"], out["attr"], out["dist"], out["sal"] = p.numpy(), a.numpy(), d.numpy(), s.numpy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/precision_ablation.json")
    ap.add_argument("--oracle", type=int, default=2, help="inputs per model / size also run through the CPU oracle (0 = skip)")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--n", type=int, default=12, help="inputs per model")
    ap.add_argument("--policies", default=",".join(POLICIES))
    args = ap.parse_args()
    topiq_sets = [synthetic_images(41, args.n, 512, 512), synthetic_images(42, 4, 1024, 1024)]
    rng = np.random.default_rng(7)
    clip_x = rng.normal(0, 1, (args.n, 3, 224, 224)).astype(np.float32)
    samp_x = rng.normal(0, 1, (args.n, 3, 224, 224)).astype(np.float32)
    res, outs = {}, {}
    for pol in args.policies.split(","):
        t0 = time.time()
        try:
            outs[pol] = run_policy(pol, args.seed, topiq_sets, clip_x, samp_x, timing=True)
        except Exception as ex:      # a policy this build does not have: say so and go on
            print(f"[{pol}] not run: {ex}", flush=True)
            continue
        print(f"[{pol}] ran in {time.time() - t0:.1f} s", flush=True)
    ref = outs["f32"]
    for pol, o in outs.items():
        res[pol] = {"vs_fp32_engine": errors(o, ref),
                    "ms_per_image": {k[2:-11]: round(v, 4) for k, v in o.items() if k.startswith("t_")}}
    if args.oracle > 0:
        t0 = time.time()
        orc = oracle_outputs(args.seed, topiq_sets, clip_x, samp_x, args.oracle)
        print(f"[oracle] {args.oracle} inputs per model in {time.time() - t0:.1f} s", flush=True)
        n = args.oracle
        for pol, o in outs.items():
            # align the engine's TOPIQ rows with the oracle's (first n of every size)
            idx = np.concatenate([np.arange(n), args.n + np.arange(n)])
            sub = dict(o)
            sub["topiq_raw"] = o["topiq_raw"][idx]
            oref = dict(orc)
            oref["topiq_raw"] = np.concatenate(orc["topiq_sets"])
            res[pol]["vs_cpu_oracle"] = errors(sub, oref)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump({"seed": args.seed, "n": args.n, "inputs": "TOPIQ: n x 512^2 + 4 x 1024^2 uint8 noise; CLIP / SAMP: n x N(0,1) [3,224,224]",
               "gate": "SURVEY 8(d): 1e-3 on final scores", "policies": res}, open(args.out, "w"), indent=1)
    keys = ["topiq_rel", "aesthetic_rel", "clip_one_minus_cos", "comp_score_rel", "pattern_argmax_same", "saliency_maxabs", "score_dist_maxabs"]
    for ref_name in ("vs_fp32_engine", "vs_cpu_oracle"):
        print(f"\n== {ref_name} ==")
        print(f"{'policy':10s} " + " ".join(f"{k:>20s}" for k in keys) + "   ms/img topiq clip samp")
        for pol, r in res.items():
            if ref_name not in r:
                continue
            e = r[ref_name]
            ms = r["ms_per_image"]
            print(f"{pol:10s} " + " ".join(f"{e[k]!s:>20.20s}" if isinstance(e[k], bool) else f"{e[k]:20.3e}" for k in keys) +
                  f"   {ms.get('topiq', 0):.3f} {ms.get('clip', 0):.3f} {ms.get('samp', 0):.3f}")


if __name__ == "__main__":
    main()
