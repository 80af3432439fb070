#!/usr/bin/env python3
"""Headline benchmark: images/s of the scoring hot path on synthetic 1024x1024 RGB batches.

  python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher's RANK in the environment: this process starts N fresh rank processes itself
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`, before anything here touches the GPU),
relays rank 0's JSON line and exits with the children's status. Under a launcher (RANK set) it is one rank:
one process per GPU, images sharded in contiguous blocks, one RCCL all-gather of the per-image records per step.

A "step" = one pass of the hot path over one batch resident in HBM. The default workload is the one BASELINE.json's
metric names - the TOPIQ + SAMP-Net + CLIP + InsightFace ensemble (`full`) at 1024x1024, 256 images per GPU - with
synthetic checkpoints (facet_amd.weights, seed 3), seeded stand-in ONNX graphs of the three buffalo_l architectures and
synthetic uint8 images. Per-GPU work is fixed as N grows (weak scaling); `value` = images all ranks
scored / max-over-ranks wall time.

Rank 0 prints ONE JSON line. Besides the contract's fields it carries
  roofline      fp32-MFMA bound; achieved = EXECUTED FLOPs of the timed region / HIP-event time on the engine stream
                (Winograd layers counted with the multiply-adds they really issue); `effective_tflops` is the same
                time against the direct-convolution (algorithmic) FLOP count
  cpu_baseline  the torch-CPU oracle port on a bounded sample of the same workload
  sub           at N = 1: BASELINE configs[1] (TOPIQ fp32 b256) and north_star's target line (TOPIQ+CLIP forward at
                batch 256), each timed the same way with its own roofline
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DTYPE_FIELD = {"f32": "f32", "bf16": "bf16", "f16": "f16", "reference_gpu": "f32 + f16 CLIP", "parity": "f16 TOPIQ / U2-Net-P, f16x3 CLIP (split operands), f32 SAMP-Net",
               "fast16": "f16 (fp32 residual streams)"}
METRIC = "images/sec whole-node (TOPIQ+SAMP+CLIP+InsightFace ensemble), 1024² batch"
FACES_PER_IMAGE = 2
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: bf16 MFMA dense peak (~2.5 PF; the 5 PF headline includes 2:1 sparsity)
TRAFFIC_FILE = "r03_traffic.json"   # written by tools/traffic_all.sh + tools/collect_traffic.py from this round's --pmc passes
KERNEL_NAMES = {False: "conv_dma_kernel + conv_igemm_kernel + stem_kernel (fp32 v_mfma_f32_32x32x2_f32 implicit GEMM)",
                True: "conv_bf16_kernel<E> + attn_fwd_bf16_kernel<E> + stem7_bf16_kernel (2-byte v_mfma_f32_32x32x16_{f16,bf16} implicit GEMM; the "
                      "fp32 models of the policy on conv_dma_kernel)"}
WORKLOADS = ["full", "topiq", "topiq_clip", "ensemble", "faces"]
# which models a workload runs: ensemble mask (1 topiq | 2 clip | 4 samp) and whether the face stage runs
WL = {"topiq": (1, False), "topiq_clip": (3, False), "ensemble": (7, False), "faces": (5, True), "full": (7, True)}


FACE_TEXT = (f"InsightFace-style SCRFD detect @640 + 2d106 landmarks + ArcFace-R50 ({FACES_PER_IMAGE} best faces/image, seeded "
             "stand-in ONNX graphs of the buffalo_l architectures)")
HALF_NOTE = ("2-byte models: activations / weights in that type, fp32 accumulate; fp32 stay CLIP's 14x14 patch embedding, non-7x7 first "
             "layers, LayerNorm / softmax statistics, score heads")
POLICY_NOTE = {
    "f32": "fp32",
    "bf16": f"bf16 on all three models (BASELINE configs[3] taken literally; {HALF_NOTE})",
    "f16": f"fp16 on all three models ({HALF_NOTE})",
    "reference_gpu": "the reference's own GPU precisions: CLIP in fp16 (`self.model.half()`, processing/scorer.py:513-516; aesthetic MLP fp32 "
                     "on the fp32 features as there), everything else fp32",
    "parity": "precision policy PARITY (facet_amd/precision.py): the fastest per-model assignment whose final scores stay within SURVEY 8(d)'s "
              f"1e-3 of the fp32 oracle - TOPIQ fp16 (images below 256 x 256 pixels on its fp32 weights: none in this workload), U2-Net-P fp16, SAMP-Net fp32, CLIP split-operand fp16 (every GEMM operand an fp16 pair hi + lo, "
              f"3x the matrix work of plain fp16; {HALF_NOTE})",
    "fast16": "precision policy FAST16: TOPIQ / U2-Net-P fp16, CLIP and SAMP-Net fp16 with fp32 residual streams (scores within 5e-3, "
              f"embedding cosine >= 1 - 1e-6: outside the 1e-3 gate; {HALF_NOTE})",
}


def workload_text(wl, B, HW, policy="f32"):
    """One sentence naming the workload and its arithmetic (policy: a name of facet_amd/precision.py POLICIES)."""
    prec = POLICY_NOTE[policy]
    cfg3 = ("no InsightFace; the fp32 form of BASELINE configs[3]" if policy == "f32" else
            "no InsightFace; BASELINE configs[3] = the 16gb profile at reduced precision")
    return {
        "topiq": f"TOPIQ-NR (ResNet-50 pyramid + CFANet head), batch {B}/GPU, {HW}x{HW} RGB (BASELINE.json configs[1]); {prec}",
        "faces": f"TOPIQ + SAMP-Net/U2-Net-P + {FACE_TEXT}, batch {B}/GPU, {HW}x{HW} (BASELINE.json configs[2]); {prec}; face graphs fp32",
        "full": f"whole ensemble of the metric: TOPIQ + SAMP-Net/U2-Net-P + CLIP ViT-L/14 + aesthetic MLP + {FACE_TEXT}, batch {B}/GPU, "
                f"{HW}x{HW}; {prec}; face graphs fp32",
        "topiq_clip": f"TOPIQ-NR + CLIP ViT-L/14 image tower + aesthetic MLP (north_star's 'TOPIQ+CLIP forward'), batch {B}/GPU, {HW}x{HW} RGB; {prec}",
        "ensemble": f"TOPIQ + SAMP-Net/U2-Net-P + CLIP ViT-L/14 + aesthetic MLP ({cfg3}), batch {B}/GPU, {HW}x{HW} RGB; {prec}",
    }[wl]


def cpu_baseline(sample, hw, seed_w, workload="topiq"):
    """Oracle (torch-CPU fp32 port) on a bounded sample with the reference's calling pattern: TOPIQ and SAMP-Net one
    image per forward (models/pyiqa_scorer.py:245-253, processing/multi_pass.py:540-547), CLIP batched per chunk
    (multi_pass.py:518-525); preprocessing with PIL like the reference."""
    import torch
    from facet_amd.weights import synthetic_state_dict, synthetic_images
    from oracle.topiq import CFANet
    ld = lambda net, name: (net.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(name, seed_w).items()}),
                            net.eval())[1]
    net = ld(CFANet(), "topiq")
    imgs = synthetic_images(2, sample, hw, hw)
    # the whole host, whatever share of it the launcher gave this rank's OpenMP pool (self_launch sets OMP_NUM_THREADS = cpus / N
    # for the ranks' own host glue): the baseline at N > 1 is the same measurement as at N = 1
    try:
        import psutil
        host_threads = psutil.cpu_count(logical=False) or os.cpu_count() or 1      # physical cores: torch's own default at N = 1
    except ImportError:
        host_threads = os.cpu_count() or 1
    try:
        host_threads = min(host_threads, len(os.sched_getaffinity(0)))             # never more than this process may run on
    except AttributeError:
        pass
    if torch.get_num_threads() != host_threads:      # (only a launcher-limited rank changes its pool: at N = 1 this is torch's default)
        torch.set_num_threads(max(1, host_threads))
    cores = torch.get_num_threads()
    extras = []
    if workload in ("faces", "full"):
        from PIL import Image
        from oracle.sampnet import U2NETP, SAMPNet
        from oracle import face_ref
        from standins import synthetic_onnx as SO
        u2, sn = ld(U2NETP(), "u2netp"), ld(SAMPNet(), "samp_net")
        im_m, im_s = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        fm = {"det": (SO.scrfd_like(seed=12, size=640)[0], 127.5, 128.0), "lmk": (SO.landmark_like(seed=13)[0], 0.0, 1.0),
              "rec": (SO.arcface_iresnet(seed=14)[0], 127.5, 127.5)}

        def extra_faces(batch):
            for a in batch:      # per image, like analyzers/face.py:99 and multi_pass.py:540-547
                if workload == "faces":
                    x = torch.from_numpy(np.asarray(Image.fromarray(a).resize((224, 224), Image.BILINEAR), np.float32) / 255).permute(2, 0, 1)[None]
                    x = (x - im_m) / im_s
                    sn(x, u2(x))
                det, kpss = face_ref.scrfd_detect(fm["det"][0], a, (640, 640))
                for f in range(min(FACES_PER_IMAGE, det.shape[0])):
                    face_ref.landmark_get(fm["lmk"][0], a, det[f, :4], 192, 0.0, 1.0)
                    face_ref.arcface_get(fm["rec"][0], a, kpss[f], 127.5, 127.5)
        extras.append(extra_faces)
    if workload in ("ensemble", "full", "topiq_clip"):
        from PIL import Image
        from oracle.clip_vit import CLIPImage, aesthetic_head, CLIP_MEAN, CLIP_STD
        from oracle.sampnet import U2NETP, SAMPNet
        clip, head, u2, sn = ld(CLIPImage(), "clip"), ld(aesthetic_head(), "aesthetic"), ld(U2NETP(), "u2netp"), ld(SAMPNet(), "samp_net")
        im_m, im_s = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        cl_m, cl_s = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1), torch.tensor(CLIP_STD).view(1, 3, 1, 1)

        def extra_ens(batch):
            pils = [Image.fromarray(a) for a in batch]
            c_in = torch.stack([torch.from_numpy(np.asarray(p.resize((224, 224), Image.BICUBIC), np.float32) / 255).permute(2, 0, 1)
                                for p in pils])
            f = clip.encode_image((c_in - cl_m) / cl_s)
            head(f)
            for p in (pils if workload != "topiq_clip" else []):
                x = torch.from_numpy(np.asarray(p.resize((224, 224), Image.BILINEAR), np.float32) / 255).permute(2, 0, 1)[None]
                x = (x - im_m) / im_s
                sn(x, u2(x))
        extras.append(extra_ens)
    with torch.no_grad():
        x = torch.from_numpy(imgs[:1].astype(np.float32) / 255.0).permute(0, 3, 1, 2)
        net(x)  # warm
        t0 = time.perf_counter()
        for i in range(sample):
            x = torch.from_numpy(imgs[i:i + 1].astype(np.float32) / 255.0).permute(0, 3, 1, 2)
            net(x)
        for ex in extras:
            ex(imgs)
        dt = time.perf_counter() - t0
    what = {"topiq": "oracle/topiq.py CFANet", "topiq_clip": "oracle TOPIQ + CLIP ViT-L/14 + aesthetic MLP", "ensemble": "oracle TOPIQ + CLIP ViT-L/14 + aesthetic MLP + U2NETP + SAMPNet",
            "faces": f"oracle TOPIQ + U2NETP + SAMPNet + face_ref SCRFD@640/landmarks/ArcFace ({FACES_PER_IMAGE} faces/image)",
            "full": f"oracle TOPIQ + CLIP ViT-L/14 + aesthetic MLP + U2NETP + SAMPNet + face_ref SCRFD@640/landmarks/ArcFace "
                    f"({FACES_PER_IMAGE} faces/image)"}[workload]
    cpu_model = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), cpu_model)
    except OSError:
        pass
    return {"value": round(sample / dt, 4), "unit": "images/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
            "sample": f"{sample} x {hw}x{hw} synthetic RGB through {what} (torch {torch.__version__} CPU fp32, {cores} "
                      f"threads on {cpu_model}; TOPIQ/SAMP one image per forward, CLIP one batch)"}


def vlm_sub(dev_index, progress, layers=4):
    """BASELINE configs[4] (24gb profile: Qwen2.5-VL-7B tagger, models/vlm_tagger.py): the engine's vision tower + text decoder at the 7B
    geometry on seeded weights. The FULL 32-block vision tower; `layers` of the 28 decoder layers plus the whole 152064-row lm_head (the
    per-layer work is what is measured; drawing 7.6 G synthetic parameters would take minutes - the 28-layer figures are extrapolated
    from the per-layer time and say so). bf16, as the reference loads the model. Units: tokens/s and images/s, not the metric's."""
    from facet_amd import Engine
    from facet_amd._lib import FE_MODEL_VLM
    from facet_amd.weights import synthetic_state_dict, qwen2_5_vl_text_spec, qwen2_5_vl_vision_spec
    from facet_amd.vlm_tagger import vision_indices
    H, NH, NKV, INTER, V = 3584, 28, 4, 18944, 152064
    sd = synthetic_state_dict(None, 3, spec=qwen2_5_vl_text_spec(hidden=H, layers=layers, heads=NH, kv_heads=NKV, inter=INTER, vocab=V) + qwen2_5_vl_vision_spec())
    e = Engine(dev_index, arena_bytes=40 << 30)
    try:
        e.vlm_configure(NH, NKV, 128, 1e6, 1e-6, (16, 24, 24))
        e.vlm_vision_configure(16, (7, 15, 23, 31))
        e.load_weights(FE_MODEL_VLM, sd)
        del sd
        progress("vlm context ready (Qwen2.5-VL-7B geometry)")
        layer_p = H * (NH + 2 * NKV) * 128 + NH * 128 * H + 3 * H * INTER
        full_scale = lambda ms_l, ms_head: ms_l * 28 / layers + ms_head
        out = {"model": f"Qwen2.5-VL-7B geometry, seeded weights, bf16; vision tower 32 blocks; {layers} of 28 decoder layers + full lm_head measured",
               "reference": "models/vlm_tagger.py:163-184, 245-259 (transformers Qwen2_5_VLForConditionalGeneration, generate(do_sample=False))"}
        n_img = 8
        idx = vision_indices([[1, 74, 74]] * n_img)
        pv = np.random.default_rng(0).normal(0, 1, (74 * 74 * n_img, 1176)).astype(np.float32)
        e.vlm_encode_images(pv, idx["patch_pos_hw"], idx["window_index"], idx["cu_window_seqlens"], idx["cu_seqlens"], want_embeds=False)
        e.timer_start()
        e.vlm_encode_images(pv, idx["patch_pos_hw"], idx["window_index"], idx["cu_window_seqlens"], idx["cu_seqlens"], want_embeds=False)
        ms = e.timer_stop()
        out["vision_tower"] = {"images_per_s": round(n_img / ms * 1e3, 1), "image": "1036x1036 px = 74x74 patches -> 1369 image tokens", "batch": n_img,
                               "note": "host patches uploaded inside the call"}
        steps = 16
        for B in (2, 32):      # the reference's vlm_batch_size, and configs[4]'s batch
            L = 512
            p = np.random.default_rng(B).integers(0, V, (B, L)).astype(np.int32)
            e.vlm_prefill(p, max_seq=L + steps + 8)
            e.timer_start(); nxt = e.vlm_prefill(p, max_seq=L + steps + 8); t_pre = e.timer_stop()
            pos = np.full((3, B), L, np.int32)
            e.vlm_decode_step(nxt, pos)
            t0 = time.perf_counter()
            for s_ in range(steps):
                nxt = e.vlm_decode_step(nxt, pos + 1 + s_)
            t_dec = (time.perf_counter() - t0) / steps * 1e3
            wbytes = 2.0 * (layers * layer_p + V * H)
            w_full = 2.0 * (28 * layer_p + V * H)
            t_full = t_dec * w_full / wbytes
            out[f"batch_{B}"] = {"prefill_tokens_per_s": round(B * L / t_pre * 1e3), "prefill_tflops": round(2.0 * layers * layer_p * B * L / t_pre / 1e9, 1),
                                 "decode_ms_per_step_measured": round(t_dec, 3), "decode_weight_GBps": round(wbytes / t_dec / 1e6),
                                 "decode_frac_of_hbm_peak": round(wbytes / t_dec / 1e6 / 8000, 3),
                                 "decode_ms_per_step_28_layers_extrapolated": round(t_full, 2),
                                 "decode_tokens_per_s_28_layers_extrapolated": round(B / t_full * 1e3)}
        # the fields every sub line carries: value = decode tokens/s at configs[4]'s batch (measured layers only: NOT the 28-layer model);
        # roofline of the dominant kernel of a decode step - the weight stream (vlm_gemm32_kernel / vlm_gemv_kernel), HBM-bound
        b32 = out["batch_32"]
        out["value"] = round(32 / b32["decode_ms_per_step_measured"] * 1e3, 1)
        out["unit"] = f"decode tokens/s at batch 32 through {layers} of 28 decoder layers + the full lm_head"
        out["dtype"] = "bf16"
        out["config"] = {"workload": "Qwen2.5-VL-7B geometry (BASELINE configs[4], models/vlm_tagger.py): 32 sequences x 512 prompt tokens, greedy decode; seeded weights; "
                                     f"{layers} of 28 decoder layers + lm_head (152064 x 3584) measured, the 28-layer figures beside them are extrapolated",
                         "batch": 32, "prompt_tokens": 512}
        out["roofline"] = {"bound": "hbm", "achieved": b32["decode_weight_GBps"], "peak": 8000.0, "unit": "GB/s", "frac": round(b32["decode_weight_GBps"] / 8000.0, 4),
                           "traffic": None, "kernel": "vlm_gemm32_kernel (weight-streaming v_mfma_f32_32x32x16_bf16, 5..32 sequences) / vlm_gemv_kernel (<= 4)",
                           "note": "achieved = bf16 weight bytes of the measured layers + lm_head per decode step / step time (whole step, all launches)"}
        progress(f"sub.vlm_tagger: vision {out['vision_tower']['images_per_s']} images/s, decode {out['batch_32']['decode_ms_per_step_measured']} ms/step at batch 32")
        return out
    finally:
        e.close()


def self_launch(args, argv):
    """`python bench.py --gpus N` typed directly: start N fresh rank processes. Runs before this process imports torch or the
    engine, so no process that touched the GPU is ever replaced or re-executed; the children's stdout (rank 0's JSON line) passes
    through and their exit status becomes ours."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    return subprocess.run(cmd, env=env).returncode


def traffic_bytes(workload, HW, B):
    """HBM traffic of the workload from the PMC passes (profiles/rNN_traffic.json; collected with rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate runs of this script; gfx950 correction: FETCH_SIZE counts 64 B per 128-B request, so it is doubled -
    /opt/skills/guides/MI355X_MICROARCH.md, HBM section). Bytes per STEP of one GPU, or None when that workload was not collected."""
    name = TRAFFIC_FILE      # this round's passes only: bytes measured on older kernels are never attached to this round's timings
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        tj = json.load(open(path)).get(workload)
        if tj and tj.get("image_size") == HW:
            return int((2.0 * tj["fetch_kb_per_image"] + tj["write_kb_per_image"]) * 1024 * B), name
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--microbatch", type=int, default=32)
    ap.add_argument("--workload", choices=WORKLOADS, default="full",
                    help="full = the metric's whole ensemble (TOPIQ + SAMP + CLIP + InsightFace-style faces); topiq = BASELINE.json configs[1]; "
                         "topiq_clip = the 'TOPIQ+CLIP forward' of north_star's roofline target; ensemble = TOPIQ + SAMP-Net/U2-Net-P + CLIP "
                         "ViT-L/14 + aesthetic MLP; faces = configs[2]: TOPIQ + SAMP-Net + SCRFD/landmarks/ArcFace")
    ap.add_argument("--cpu-sample", type=int, default=2, help="images for the CPU baseline leg (0 = skip)")
    ap.add_argument("--no-sub", action="store_true", help="skip the configs[1] / TOPIQ+CLIP sub-measurements")
    ap.add_argument("--dtype", choices=list(POLICY_NOTE), default="f32",
                    help="precision policy of TOPIQ / SAMP-Net / U2-Net-P / CLIP for the primary workload (facet_amd/precision.py; face graphs "
                         "always fp32); the headline stays f32 = the reference CPU path's arithmetic")
    ap.add_argument("--dry-run", action="store_true", help="launch / rendezvous / gather path only, no engine (CPU test hook, gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # the host driver of this pool only supports dmabuf IPC: RCCL needs this before the first HIP call (it is exported on the
    # GPU boxes already; set here too so a bare environment cannot break the N>1 run)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("FACET_AMD_SYNTHETIC", "1")   # no checkpoint files offline: seeded synthetic weights, stated in `config`
    import torch
    import torch.distributed as dist
    if args.dry_run:
        # the N > 1 control flow without a GPU: every rank gathers one record per image of its shard over gloo
        from facet_amd.sharding import shard_range, gather_scores
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo")
        lo, hi = shard_range(4 * world, world, rank)
        full = gather_scores(np.arange(lo, hi, dtype=np.float32), world)
        ok = bool(np.array_equal(full, np.arange(4 * world, dtype=np.float32)))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "gathered_ok": ok}), flush=True)
        sys.exit(0 if ok else 1)
    # Rehearsal knobs (not used by the driver): FACET_BENCH_BACKEND=gloo + FACET_BENCH_DEVICE=0 let two ranks share one
    # GPU so the N>1 control flow can be exercised on a 1-GPU box; the real multi-GPU run is nccl (= RCCL over xGMI).
    backend = os.environ.get("FACET_BENCH_BACKEND", "nccl")
    dev_index = int(os.environ.get("FACET_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    from facet_amd import Engine
    from facet_amd._lib import (FE_MODEL_TOPIQ, FE_MODEL_CLIP, FE_MODEL_AESTHETIC, FE_MODEL_SAMP, FE_MODEL_U2NETP)
    from facet_amd.weights import synthetic_state_dict
    from facet_amd.sharding import shard_range, gather_scores, score_shard
    from facet_amd import precision as precision_mod

    B, HW = args.batch, args.size
    primary = args.workload
    subs = [] if (args.no_sub or world > 1) else [w for w in ("topiq", "topiq_clip") if w != primary]
    need = set([primary] + subs)
    need_mask = 0
    for w in need:
        need_mask |= WL[w][0]
    need_faces = any(WL[w][1] for w in need)

    def half_share(policy, mask):
        """Does the policy put the 1024-square model (TOPIQ) on 2-byte activations? Then the same arena holds twice the images."""
        return precision_mod.resolve(policy)["topiq"] != "f32" and (mask & 1)

    def mb_of(policy, mask=7):
        return args.microbatch * 2 if half_share(policy, mask) else args.microbatch

    sd_cache = {}

    def sd_of(name):      # the seeded checkpoints are drawn once per process, whatever number of contexts loads them
        if name not in sd_cache:
            sd_cache[name] = synthetic_state_dict(name, seed=3)
        return sd_cache[name]

    t_start = time.time()

    def progress(msg):      # stderr: the one JSON line stays alone on stdout; a run that prints nothing for minutes looks hung
        if rank == 0:
            print(f"[bench +{time.time() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    def make_engine(policy, mask):
        e = Engine(dev_index, arena_bytes=(4 + 2 * args.microbatch * max(1, (HW * HW) // (1024 * 1024))) << 30)
        names = ["topiq"] + (["clip", "aesthetic"] if mask & 2 else []) + (["u2netp", "samp_net"] if mask & 4 else [])
        precision_mod.load_models(e, policy, {n: sd_of(n) for n in names})      # precision is a property of a model's committed weights
        progress(f"context ready: {precision_mod.describe(policy)}")
        # 2-byte activations are half the bytes: the same arena holds twice the images per micro-batch (fewer, larger launches)
        e.set_microbatch(mb_of(policy, mask))
        return e

    def make_face_engine():
        # BASELINE.json configs[2] / the metric's InsightFace stage. Seeded stand-in graphs of the buffalo_l architectures
        # (no model files offline); uniform-noise images carry no real faces, so the best FACES_PER_IMAGE detections of the
        # synthetic detector go through landmarks + ArcFace (fixed faces-per-image mode). The stage owns a
        # context (stream + arena) of its own: its host glue (NMS, alignment matrices) sits between its launches, so
        # score_shard runs it in a worker thread beside the ensemble's launches, as BatchScorer does with aux_engine.
        from standins import synthetic_onnx as SO
        from facet_amd._lib import FE_GRAPH_FACE_DET, FE_GRAPH_FACE_LMK, FE_GRAPH_FACE_REC
        e = Engine(dev_index, arena_bytes=(4 + args.microbatch // 2) << 30)
        e.graph_load(FE_GRAPH_FACE_DET, SO.scrfd_like(seed=12, size=640)[0])
        e.graph_load(FE_GRAPH_FACE_LMK, SO.landmark_like(seed=13)[0])
        e.graph_load(FE_GRAPH_FACE_REC, SO.arcface_iresnet(seed=14)[0])
        e.set_microbatch(args.microbatch)
        return e

    eng = make_engine(args.dtype, need_mask)
    face_eng = make_face_engine() if need_faces else None

    # this rank's shard of the global batch (weak scaling: B images per GPU), generated once, resident in HBM
    lo, hi = shard_range(B * world, world, rank)
    assert hi - lo == B
    rng = np.random.default_rng([2, rank])
    d_imgs = eng.dev_alloc(B * HW * HW * 3)
    chunk = 32
    import ctypes
    for i in range(0, B, chunk):
        nb = min(chunk, B - i)
        a = rng.integers(0, 256, (nb, HW, HW, 3), dtype=np.uint8)
        eng.h2d(ctypes.c_void_p(d_imgs.value + i * HW * HW * 3), a)
    images = (d_imgs, B, HW, HW)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def make_step(wl, eng):
        mask, faces = WL[wl]
        if wl == "topiq":
            return lambda: gather_scores(eng.topiq_score(images), world, dev_index)
        fargs = ((640, 640), 0.5, 0.4, FACES_PER_IMAGE) if faces else None      # noise serves as BGR for the face stage

        def step():
            # records are written into a device buffer, all-gathered in place (RCCL) and copied to the host once
            rec, ran = score_shard(eng, images, B * world, world, rank, faces=fargs, face_engine=face_eng if faces else None)
            assert ran == mask, (ran, mask)
            return rec
        return step

    def measure(wl, steps, warmup, eng):
        eng.ensemble_select(WL[wl][0])
        step = make_step(wl, eng)
        for _ in range(warmup):
            step()
        eng.flops_reset()
        if face_eng is not None:
            face_eng.flops_reset()
        barrier()
        eng.timer_start()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        barrier()
        dt = time.perf_counter() - t0
        ev_ms = eng.timer_stop()
        assert out.shape[0] == B * world and np.isfinite(out).all()
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        fl, fx, fh = eng.flops(), eng.flops_executed(), eng.flops_half()
        if face_eng is not None and WL[wl][1]:
            fl, fx = fl + face_eng.flops(), fx + face_eng.flops_executed()      # the face graphs' contractions run on the other context
        return float(t.item()), ev_ms, fl, fx, fh

    def per_launch(wl, eng, policy):
        """Per-launch view of the dominant kernel family (rank 0, outside the timed region): one micro-batch with a HIP event pair
        around every contraction launch (engine profile mode; the per-launch sync makes it slightly pessimistic). The averages are
        what `rocprofv3 --kernel-trace --stats` reports for conv_dma_kernel (profiles/)."""
        nb = min(B, mb_of(policy, WL[wl][0]))
        eng.ensemble_select(WL[wl][0])
        eng.profile_enable(True)
        if wl == "topiq":
            eng.topiq_score((d_imgs, nb, HW, HW))
        else:
            eng.ensemble_score((d_imgs, nb, HW, HW))
        recs = [r for r in eng.profile_records() if r["ms"] > 0]
        eng.profile_enable(False)
        if not recs:
            return None
        tot_ms = sum(r["ms"] for r in recs)
        tot_fl = sum(r["flops"] for r in recs)
        return {"launches": len(recs), "avg_us": round(tot_ms / len(recs) * 1e3, 1), "images": nb,
                "effective_tflops": round(tot_fl / tot_ms / 1e9, 2),
                "note": f"contraction launches only ({KERNEL_NAMES[policy != 'f32']}; face graphs excluded), "
                        "algorithmic FLOPs / summed launch durations of one micro-batch"}

    def roofline(wl, steps, ev_ms, flops, flops_exec, flops_half, eng, policy):
        """bound = mfma. One dtype: achieved / peak of that dtype. A policy that mixes fp32 and 2-byte models shares one launch
        stream: frac = (fp32 FLOPs / fp32 MFMA peak + 2-byte FLOPs / 2-byte MFMA peak) / measured time, i.e. the time the matrix pipes
        would need at peak over the time taken; the engine counts the 2-byte share itself (fe_flops_get_half)."""
        t = ev_ms * 1e-3
        achieved = flops_exec / t / 1e12
        traffic, tfile = traffic_bytes(wl + ("" if policy == "f32" else "_" + policy), HW, B)
        f16_fl = min(flops_half, flops_exec)
        f32_fl = flops_exec - f16_fl
        mixed = f16_fl > 0.01 * flops_exec and f32_fl > 0.01 * flops_exec
        if mixed:
            peak = {"f32": FP32_MFMA_PEAK_TFLOPS, "2-byte": BF16_MFMA_PEAK_TFLOPS}
            frac = (f32_fl / (FP32_MFMA_PEAK_TFLOPS * 1e12) + f16_fl / (BF16_MFMA_PEAK_TFLOPS * 1e12)) / t
        else:
            peak = BF16_MFMA_PEAK_TFLOPS if f16_fl > 0.5 * flops_exec else FP32_MFMA_PEAK_TFLOPS
            frac = achieved / peak
        r = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
             "frac": round(frac, 4), "traffic": traffic,
             "traffic_unit": f"bytes per step per GPU (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/{tfile})" if tfile else None,
             "kernel": KERNEL_NAMES[policy != "f32"] + "; every contraction launched in the timed region",
             "effective_tflops": round(flops / t / 1e12, 2),
             "flops_per_image": round(flops / (B * steps), 1),
             "executed_flops_per_image": round(flops_exec / (B * steps), 1),
             "note": "achieved / frac count the FLOPs the matrix cores executed: fp32 3x3 stride-1 pad-1 convs with >= 96 input channels run as "
                     "Winograd F(4x4,3x3) (36 batched GEMMs per launch, 4x fewer multiply-adds); effective_tflops divides the direct "
                     "convolution's (algorithmic) FLOPs by the same time",
             "event_ms": round(ev_ms, 3), "per_launch": per_launch(wl, eng, policy) if rank == 0 else None}
        if mixed:
            r["executed_flops_per_image_by_dtype"] = {"f32": round(f32_fl / (B * steps), 1), "2-byte": round(f16_fl / (B * steps), 1)}
            r["note"] += "; two dtypes share the launch stream: frac = (fp32 FLOPs / fp32 MFMA peak + 2-byte FLOPs / 2-byte MFMA peak) / measured time"
        return r

    dt_max, ev_ms, flops, flops_exec, flops_half = measure(primary, args.steps, args.warmup, eng)
    progress(f"{primary} [{args.dtype}]: {B * world * args.steps / dt_max:.1f} images/s")
    overrides = {k: v for k, v in os.environ.items() if k.startswith("FE_") or k == "FACET_AMD_LIB"}      # tuning hooks that change the measured path
    out = None
    if rank == 0:
        total_images = B * world * args.steps
        out = {
            "metric": METRIC, "value": round(total_images / dt_max, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE_FIELD[args.dtype], "data": "synthetic",
            "config": {"workload": workload_text(primary, B, HW, args.dtype),
                       "precision": precision_mod.describe(args.dtype),
                       "global_batch": B * world, "image_size": HW, "microbatch": mb_of(args.dtype, WL[primary][0]),
                       "parallelism": f"image-sharded x{world}, one RCCL all-gather of per-image records per step (device buffers)",
                       "weights": "seeded synthetic checkpoints (no weight files offline)",
                       "env_overrides": overrides or None},
            "roofline": roofline(primary, args.steps, ev_ms, flops, flops_exec, flops_half, eng, args.dtype),
        }
    # sub-measurements (N = 1 only, f32 headline only): BASELINE configs[1], north_star's TOPIQ+CLIP target line (>= 10 timed steps), and the
    # metric's / configs[3]'s workload under the other precision policies - one spare context at a time (each holds tens of GB)
    sub = {}
    sub_runs = [(wl, wl, args.dtype, 10 if wl == "topiq_clip" else 3) for wl in subs]
    if not (args.no_sub or world > 1) and args.dtype == "f32":
        full = primary == "full"
        sub_runs += [("ensemble_parity", "ensemble", "parity", 3)]            # configs[3] under the policy that holds the 1e-3 gate
        if full:
            sub_runs += [("full_parity", "full", "parity", 3),                # the metric's workload under that policy
                         ("full_clip_f16", "full", "reference_gpu", 3)]       # the reference's own GPU precisions: CLIP halved, the rest fp32
        sub_runs += [("ensemble_fast16", "ensemble", "fast16", 3),            # every model in fp16 (fp32 residual streams)
                     ("ensemble_bf16", "ensemble", "bf16", 3)]                # BASELINE configs[3] taken literally
    spare, spare_pol = None, None      # one spare context at a time, shared by consecutive runs of one policy
    for key, wl, pol, s_steps in sub_runs:
        s_warm = 1
        e = eng
        if pol != args.dtype:
            if spare_pol != pol:
                if spare is not None:
                    spare.close()
                spare, spare_pol = make_engine(pol, 7), pol
            e = spare
        dts, evs, fl, fx, fh = measure(wl, s_steps, s_warm, e)
        progress(f"sub.{key}: {B * s_steps / dts:.1f} images/s")
        sub[key] = {"value": round(B * s_steps / dts, 2), "unit": "images/s", "steps": s_steps, "warmup": s_warm,
                    "ms_per_step": round(dts / s_steps * 1e3, 3), "dtype": DTYPE_FIELD[pol],
                    "config": {"workload": workload_text(wl, B, HW, pol), "precision": precision_mod.describe(pol),
                               "microbatch": mb_of(pol, WL[wl][0])},
                    "roofline": roofline(wl, s_steps, evs, fl, fx, fh, e, pol)}
    if spare is not None:
        spare.close()
    if rank == 0 and not (args.no_sub or world > 1) and args.dtype == "f32" and primary == "full":
        sub["vlm_tagger"] = vlm_sub(dev_index, progress)      # BASELINE configs[4]'s model: its own metric (tokens/s), not images/s
    if rank == 0:
        if sub:
            out["sub"] = sub
        if args.cpu_sample > 0:
            progress("cpu_baseline ...")
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, HW, 3, primary)
            progress(f"cpu_baseline: {out['cpu_baseline']['value']} images/s on {out['cpu_baseline']['cores']} threads")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.dev_free(d_imgs)
    for e in [eng] + ([face_eng] if face_eng is not None else []):
        e.close()


if __name__ == "__main__":
    main()
