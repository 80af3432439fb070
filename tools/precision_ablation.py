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
POLICIES = ["f32", "bf16", "f16", "bf16+r32", "f16+r32", "f16x3"]


def comp_score(dist):
    raw = (dist * np.arange(1, 6, dtype=np.float64)).sum(1)
    return np.clip((raw - 1.0) / 4.0 * 10.0, 0.0, 10.0)


def _engine(loads):
    """loads: [(precision, [model names])] committed in that order on one context."""
    e = Engine(0, arena_bytes=40 << 30)
    for prec, names in loads:
        e.set_precision(prec)
        for n in names:
            e.load_weights(IDS[n], synthetic_state_dict(n, SEED))
    e.set_precision("f32")
    return e


def _time(e, fn):
    fn()
    e.timer_start()
    fn()
    return e.timer_stop()


def run_topiq(prec, topiq_sets):
    e = _engine([(prec, ["topiq"])])
    try:
        e.set_microbatch(4)
        out = {"topiq_raw": np.concatenate([e.topiq_score(imgs) for imgs in topiq_sets])}
        e.set_microbatch(8)
        out["ms"] = _time(e, lambda: e.topiq_score(topiq_sets[-1])) / len(topiq_sets[-1])
    finally:
        e.close()
    return out


def run_clip(prec, clip_x):
    e = _engine([(prec, ["clip", "aesthetic"])])
    try:
        feat, emb, aes = e.clip_encode_image(clip_x, normalized=True, aesthetic=True)
        out = {"clip_emb": emb, "clip_feat": feat, "aes_raw": aes}
        out["ms"] = _time(e, lambda: e.clip_encode_image(clip_x, normalized=True, aesthetic=True)) / len(clip_x)
    finally:
        e.close()
    return out


def run_samp(prec_u2, prec_samp, samp_x):
    e = _engine([(prec_u2, ["u2netp"]), (prec_samp, ["samp_net"])])
    try:
        pw, at, dist, sal = e.samp_forward(samp_x, want_saliency=True)
        out = {"pw": pw, "attr": at, "dist": dist, "sal": sal}
        out["ms"] = _time(e, lambda: e.samp_forward(samp_x)) / len(samp_x)
    finally:
        e.close()
    return out


def errors(got, ref):
    """Final-score errors of one run against a reference dict with the same keys (the reference may cover only the first rows)."""
    n = lambda k: min(len(got[k]), len(ref[k]))
    r = {}
    if "topiq_raw" in got:
        k = n("topiq_raw")
        # the stored score is clamp(raw, 0, 1) * 10; the error is taken on raw (the clamp would hide it), relative with the tests' floor
        r["topiq_rel"] = float((np.abs(got["topiq_raw"][:k] - ref["topiq_raw"][:k]) / np.maximum(np.abs(ref["topiq_raw"][:k]), 1e-3)).max())
    if "aes_raw" in got:
        k = n("aes_raw")
        ga, ra = (got["aes_raw"][:k] + 1) * 5, (ref["aes_raw"][:k] + 1) * 5
        r["aesthetic_rel"] = float((np.abs(ga - ra) / np.maximum(np.abs(ra), 1.0)).max())
        r["clip_one_minus_cos"] = float((1.0 - (got["clip_emb"][:k].astype(np.float64) * ref["clip_emb"][:k]).sum(1)).max())
        r["clip_emb_maxabs"] = float(np.abs(got["clip_emb"][:k] - ref["clip_emb"][:k]).max())
        r["clip_feat_rel"] = float(np.abs(got["clip_feat"][:k] - ref["clip_feat"][:k]).max() / np.abs(ref["clip_feat"][:k]).max())
    if "dist" in got:
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


SEED = 3
SAMP_POLICIES = [("f32", "f32"), ("bf16", "bf16"), ("f16", "f16"), ("bf16+r32", "bf16+r32"), ("f16+r32", "f16+r32"), ("f32", "f16"),
                 ("f32", "f16+r32"), ("f16", "f32")]


def main():
    global SEED
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/precision_ablation.json")
    ap.add_argument("--oracle", type=int, default=2, help="inputs per model / size also run through the CPU oracle (0 = skip)")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--n", type=int, default=12, help="inputs per model")
    ap.add_argument("--policies", default=",".join(POLICIES))
    args = ap.parse_args()
    SEED = args.seed
    topiq_sets = [synthetic_images(41, args.n, 512, 512), synthetic_images(42, 4, 1024, 1024)]
    rng = np.random.default_rng(7)
    clip_x = rng.normal(0, 1, (args.n, 3, 224, 224)).astype(np.float32)
    samp_x = rng.normal(0, 1, (args.n, 3, 224, 224)).astype(np.float32)
    pols = args.policies.split(",")
    runs = {"topiq": {}, "clip": {}, "samp": {}}

    def attempt(group, name, fn):
        try:
            runs[group][name] = fn()
        except Exception as ex:      # a policy this build does not have: say so and go on
            print(f"[{group} {name}] not run: {ex}", flush=True)
    for pol in pols:
        if pol != "f16x3":      # split operands exist for the ViT tower only
            attempt("topiq", pol, lambda: run_topiq(pol, topiq_sets))
        attempt("clip", pol, lambda: run_clip(pol, clip_x))
    for pu, ps in SAMP_POLICIES:
        if pu.split("+")[0] in [p.split("+")[0] for p in pols] or pu == "f32":
            attempt("samp", f"u2netp {pu} / samp_net {ps}", lambda: run_samp(pu, ps, samp_x))
    orc = None
    if args.oracle > 0:
        t0 = time.time()
        orc = oracle_outputs(args.seed, topiq_sets, clip_x, samp_x, args.oracle)
        orc["topiq_raw"] = np.concatenate(orc["topiq_sets"])
        print(f"[oracle] {args.oracle} inputs per model in {time.time() - t0:.1f} s", flush=True)
    res = {}
    refname = {"topiq": "f32", "clip": "f32", "samp": "u2netp f32 / samp_net f32"}
    for group, rs in runs.items():
        res[group] = {}
        for name, o in rs.items():
            r = {"vs_fp32_engine": errors(o, rs[refname[group]]), "ms_per_image": round(o["ms"], 4)}
            if orc is not None:
                sub = dict(o)
                if group == "topiq":      # the oracle ran the first n images of every size
                    sub["topiq_raw"] = o["topiq_raw"][np.concatenate([np.arange(args.oracle), args.n + np.arange(args.oracle)])]
                r["vs_cpu_oracle"] = errors(sub, orc)
            res[group][name] = r
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump({"seed": args.seed, "n": args.n, "inputs": "TOPIQ: n x 512^2 + 4 x 1024^2 uint8 noise; CLIP / SAMP: n x N(0,1) [3,224,224]",
               "gate": "SURVEY 8(d): 1e-3 on final scores", "results": res}, open(args.out, "w"), indent=1)
    cols = {"topiq": ["topiq_rel"], "clip": ["aesthetic_rel", "clip_one_minus_cos", "clip_feat_rel"],
            "samp": ["comp_score_rel", "pattern_argmax_same", "saliency_maxabs", "score_dist_maxabs", "attributes_maxabs"]}
    for group in ("topiq", "clip", "samp"):
        for ref_name in ("vs_fp32_engine", "vs_cpu_oracle"):
            print(f"\n== {group}: {ref_name} ==")
            print(f"{'policy':36s} " + " ".join(f"{k:>20s}" for k in cols[group]) + "   ms/img")
            for name, r in res[group].items():
                if ref_name not in r:
                    continue
                e = r[ref_name]
                print(f"{name:36s} " + " ".join(f"{e[k]!s:>20.20s}" if isinstance(e[k], bool) else f"{e[k]:20.3e}" for k in cols[group]) +
                      f"   {r['ms_per_image']:.3f}")


if __name__ == "__main__":
    main()
