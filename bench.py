#!/usr/bin/env python3
"""Headline benchmark: images/s of the scoring hot path on synthetic 1024x1024 RGB batches.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`,
   one rank per GPU, images sharded independently, per-image scores all-gathered over RCCL each step)

A "step" = one pass of the hot path over one batch resident in HBM. The N=1 workload is BASELINE.json
configs[1]: TOPIQ (ResNet-50 pyramid + CFANet head) fp32, batch 256, 1024x1024, synthetic checkpoint
(facet_amd.weights, seed 3) and synthetic uint8 images. Per-GPU work is fixed as N grows
(weak scaling); `value` = images all ranks scored / max-over-ranks wall time.

Prints ONE JSON line (rank 0) with `roofline` (fp32-MFMA bound; achieved = algorithmic FLOPs of the timed
region / HIP-event time on the engine stream) and `cpu_baseline` (torch-CPU oracle port on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "images/sec whole-node (TOPIQ+SAMP+CLIP+InsightFace ensemble), 1024² batch"
FACES_PER_IMAGE = 2
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


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
    cores = torch.get_num_threads()
    extras = []
    if workload in ("faces", "full"):
        from PIL import Image
        from oracle.sampnet import U2NETP, SAMPNet
        from oracle import face_ref
        from facet_amd import synthetic_onnx as SO
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--microbatch", type=int, default=32)
    ap.add_argument("--workload", choices=["topiq", "topiq_clip", "ensemble", "faces", "full"], default="topiq",
                    help="topiq = BASELINE.json configs[1]; topiq_clip = the 'TOPIQ+CLIP forward' of north_star's roofline target; ensemble = TOPIQ + SAMP-Net/U2-Net-P + CLIP ViT-L/14 + aesthetic MLP; "
                         "faces = configs[2]: TOPIQ + SAMP-Net + SCRFD/landmarks/ArcFace; full = the metric's whole ensemble "
                         "(TOPIQ + SAMP + CLIP + InsightFace-style faces)")
    ap.add_argument("--cpu-sample", type=int, default=4, help="images for the CPU baseline leg (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    # the host driver of this pool only supports dmabuf IPC: RCCL needs this before the first HIP call (it is exported on the
    # GPU boxes already; set here too so a bare environment cannot break the N>1 run)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
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
    from facet_amd.sharding import shard_range, gather_scores

    B, HW = args.batch, args.size
    eng = Engine(dev_index, arena_bytes=(4 + 2 * args.microbatch * max(1, (HW * HW) // (1024 * 1024))) << 30)
    eng.load_weights(FE_MODEL_TOPIQ, synthetic_state_dict("topiq", seed=3))
    if args.workload in ("topiq_clip", "ensemble", "full"):
        eng.load_weights(FE_MODEL_CLIP, synthetic_state_dict("clip", seed=3))
        eng.load_weights(FE_MODEL_AESTHETIC, synthetic_state_dict("aesthetic", seed=3))
    if args.workload in ("ensemble", "faces", "full"):
        eng.load_weights(FE_MODEL_U2NETP, synthetic_state_dict("u2netp", seed=3))
        eng.load_weights(FE_MODEL_SAMP, synthetic_state_dict("samp_net", seed=3))
    if args.workload in ("faces", "full"):
        # BASELINE.json configs[2]: TOPIQ + SAMP-Net + InsightFace. Seeded stand-in graphs of the buffalo_l architectures
        # (no model files offline); uniform-noise images carry no real faces, so the best FACES_PER_IMAGE detections of the
        # synthetic detector go through landmarks + ArcFace (fixed faces-per-image mode).
        from facet_amd import synthetic_onnx as SO
        from facet_amd._lib import FE_GRAPH_FACE_DET, FE_GRAPH_FACE_LMK, FE_GRAPH_FACE_REC
        eng.graph_load(FE_GRAPH_FACE_DET, SO.scrfd_like(seed=12, size=640)[0])
        eng.graph_load(FE_GRAPH_FACE_LMK, SO.landmark_like(seed=13)[0])
        eng.graph_load(FE_GRAPH_FACE_REC, SO.arcface_iresnet(seed=14)[0])
    eng.set_microbatch(args.microbatch)

    # this rank's shard of the global batch (weak scaling: B images per GPU), generated once, resident in HBM
    lo, hi = shard_range(B * world, world, rank)
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

    def step():
        if args.workload in ("ensemble", "topiq_clip"):
            rec, mask = eng.ensemble_score(images)      # [B, 789] per-image records (fields of models not loaded stay 0)
            assert mask == (7 if args.workload == "ensemble" else 3)
            return gather_scores(rec, world, dev_index)
        if args.workload in ("faces", "full"):
            rec, mask = eng.ensemble_score(images)      # faces: TOPIQ + SAMP fields only (CLIP not loaded)
            assert mask == (5 if args.workload == "faces" else 7)
            faces, counts, fmask = eng.face_analyze(images, (640, 640), 0.5, 0.4, FACES_PER_IMAGE)   # noise serves as BGR
            assert fmask == 7
            rec = np.concatenate([rec, counts[:, None].astype(np.float32), faces.reshape(B, -1)], axis=1)
            return gather_scores(rec, world, dev_index)
        scores = eng.topiq_score(images)
        return gather_scores(scores, world, dev_index)

    for _ in range(args.warmup):
        step()
    eng.flops_reset()
    barrier()
    eng.timer_start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        allscores = step()
    barrier()
    dt = time.perf_counter() - t0
    ev_ms = eng.timer_stop()
    flops = eng.flops()
    flops_exec = eng.flops_executed()
    assert allscores.shape[0] == B * world and np.isfinite(allscores).all()

    t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    # Per-launch view of the dominant kernel family (rank 0, after the timed region): one micro-batch with a HIP event pair
    # around every contraction launch (engine profile mode; the per-launch sync makes it slightly pessimistic). The averages
    # are what `rocprofv3 --kernel-trace --stats` reports for conv_dma_kernel (profiles/r01_kernel_stats_bench_b64_final.csv).
    per_launch = None
    if rank == 0 and args.workload == "topiq":
        nb = min(B, args.microbatch)
        eng.profile_enable(True)
        eng.topiq_score((d_imgs, nb, HW, HW))
        recs = [r for r in eng.profile_records() if r["ms"] > 0]
        eng.profile_enable(False)
        if recs:
            tot_ms = sum(r["ms"] for r in recs)
            tot_fl = sum(r["flops"] for r in recs)
            per_launch = {"launches": len(recs), "avg_us": round(tot_ms / len(recs) * 1e3, 1), "images": nb,
                          "achieved": round(tot_fl / tot_ms / 1e9, 2), "frac": round(tot_fl / tot_ms / 1e9 / FP32_MFMA_PEAK_TFLOPS, 4),
                          "note": "contraction launches only (conv_dma_kernel / conv_igemm_kernel / stem_kernel), algorithmic FLOPs / summed launch durations"}

    if rank == 0:
        # HBM traffic of the same workload from the PMC passes (profiles/r01_traffic.json; collected with rocprofv3 --pmc
        # FETCH_SIZE and --pmc WRITE_SIZE in separate runs of this script, gfx950 correction: FETCH_SIZE counts 64 B per
        # 128-B request, so it is doubled - /opt/skills/guides/MI355X_MICROARCH.md, HBM section). Bytes per STEP of one GPU.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath)).get(args.workload)
            if tj and tj.get("image_size") == HW:
                traffic = int((2.0 * tj["fetch_kb_per_image"] + tj["write_kb_per_image"]) * 1024 * B)
        total_images = B * world * args.steps
        achieved = flops / (ev_ms * 1e-3) / 1e12
        out = {
            "metric": METRIC, "value": round(total_images / dt_max, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"TOPIQ-NR (ResNet-50 pyramid + CFANet head) fp32, batch {B}/GPU, {HW}x{HW} RGB "
                                    "(BASELINE.json configs[1])") if args.workload == "topiq" else
                                   (f"TOPIQ + SAMP-Net/U2-Net-P + InsightFace-style SCRFD detect @640 + 2d106 landmarks + ArcFace-R50 "
                                    f"({FACES_PER_IMAGE} best faces/image, seeded stand-in ONNX graphs) fp32, batch {B}/GPU, {HW}x{HW} "
                                    "(BASELINE.json configs[2])") if args.workload == "faces" else
                                   (f"whole ensemble of the metric: TOPIQ + SAMP-Net/U2-Net-P + CLIP ViT-L/14 + aesthetic MLP + InsightFace-style "
                                    f"SCRFD@640 / 2d106 landmarks / ArcFace-R50 ({FACES_PER_IMAGE} best faces/image, stand-in ONNX graphs) fp32, "
                                    f"batch {B}/GPU, {HW}x{HW}") if args.workload == "full" else
                                   (f"TOPIQ-NR + CLIP ViT-L/14 image tower + aesthetic MLP fp32 (north_star's 'TOPIQ+CLIP forward'), "
                                    f"batch {B}/GPU, {HW}x{HW} RGB") if args.workload == "topiq_clip" else
                                   (f"ensemble TOPIQ + SAMP-Net/U2-Net-P + CLIP ViT-L/14 + aesthetic MLP fp32 (no InsightFace), "
                                    f"batch {B}/GPU, {HW}x{HW} RGB"),
                       "global_batch": B * world, "image_size": HW, "microbatch": args.microbatch,
                       "parallelism": f"image-sharded x{world}, RCCL all-gather of scores",
                       "weights": "seeded synthetic checkpoint (no weight files offline)"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "bytes per step per GPU (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_traffic.json)",
                         "kernel": "conv_dma_kernel + conv_igemm_kernel (fp32 v_mfma_f32_32x32x2_f32 implicit GEMM; every contraction launched in the timed region)",
                         "flops_per_image": round(flops / (B * args.steps), 1),
                         "executed_flops_per_image": round(flops_exec / (B * args.steps), 1),
                         "executed_note": "3x3 stride-1 pad-1 convs with >= 96 input channels run as Winograd F(4x4,3x3) (36 batched GEMMs per launch); `achieved` counts the direct convolution's FLOPs, "
                                          "the matrix cores executed executed_flops_per_image",
                         "event_ms": round(ev_ms, 3), "per_launch": per_launch},
        }
        if args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, HW, 3, args.workload)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.dev_free(d_imgs)
    eng.close()


if __name__ == "__main__":
    main()
