"""Pinning kit: turns the "parity unpinned" rows of DESIGN.md §2 into measured numbers on a machine that HAS what this build
container lacks - the third-party packages the reference runs its models through, and their weight files.

Nothing here runs in the offline build (every leg is skipped with a reason when its package or file is missing); it exists so a
maintainer with a normal Facet installation next to an MI355X can check, in one command, that the engine reproduces
  * pyiqa's topiq_nr          (reference models/pyiqa_scorer.py:197-231)           --topiq-weights  <cfanet_nr_*.pth|.safetensors>
  * open_clip ViT-L-14        (processing/scorer.py:640-673, models/model_manager.py:127-148)   --clip-weights <open_clip state_dict>
  * insightface buffalo_l     (analyzers/face.py:30-38,99 on onnxruntime)          --buffalo-dir <dir with det_10g/2d106det/w600k_r50.onnx>
  * OpenCV's pixel operators  (analyzers/technical.py, analyzers/composition.py)   (needs only `import cv2`)
on seeded synthetic images (or --images <dir> of real photos), and print max / relative differences per output. Weight files are read
with loaders that execute nothing from them (safetensors, torch.load(weights_only=True), the engine's own ONNX reader).

    python tools/pin_with_real_dependencies.py --topiq-weights ... --clip-weights ... --buffalo-dir ~/.insightface/models/buffalo_l
"""
import argparse
import glob
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def have(mod):
    try:
        return importlib.import_module(mod)
    except Exception as e:                      # noqa: BLE001 - any import failure means "leg skipped"
        print(f"  [skip] import {mod}: {type(e).__name__}: {e}")
        return None


def images(args, n, hw):
    if args.images:
        from PIL import Image
        files = sorted(glob.glob(os.path.join(args.images, "*")))[:n]
        return np.stack([np.asarray(Image.open(f).convert("RGB").resize((hw, hw), Image.BILINEAR)) for f in files])
    from facet_amd.weights import synthetic_images
    return synthetic_images(21, n, hw, hw)


def report(name, got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    d = np.abs(got - ref)
    print(f"  {name}: max |diff| {d.max():.3e}   max rel {(d / np.maximum(np.abs(ref), 1e-6)).max():.3e}   (n = {d.size})")


def leg_topiq(args, eng):
    print("TOPIQ-NR vs pyiqa")
    pyiqa, torch = have("pyiqa"), have("torch")
    if not (pyiqa and torch and args.topiq_weights):
        print("  [skip] needs pyiqa and --topiq-weights")
        return
    from facet_amd._lib import FE_MODEL_TOPIQ
    from facet_amd.pyiqa_scorer import load_checkpoint
    sd = load_checkpoint(args.topiq_weights)
    imgs = images(args, 4, 512)
    metric = pyiqa.create_metric("topiq_nr", device="cpu", as_loss=False)
    metric.net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    with torch.no_grad():
        ref = metric(torch.from_numpy(imgs.astype(np.float32) / 255.0).permute(0, 3, 1, 2)).flatten().numpy()
    # GatedConv's activations are parameter-free, so the checkpoint does not say which ones pyiqa uses: try every combination
    # (fe_topiq_configure) and name the one that reproduces pyiqa's scores
    for gate in ("gelu", "softplus", "relu"):
        for wblk in ("gelu", "relu", "softplus"):
            eng.topiq_configure(gate, wblk)
            eng.load_weights(FE_MODEL_TOPIQ, sd)
            report(f"score (gate_act={gate}, weight_blk_act={wblk})", eng.topiq_score(imgs), ref)
    eng.topiq_configure("gelu", "gelu")


def leg_clip(args, eng):
    print("CLIP ViT-L/14 image tower vs open_clip")
    oc, torch = have("open_clip"), have("torch")
    if not (oc and torch and args.clip_weights):
        print("  [skip] needs open_clip and --clip-weights")
        return
    from facet_amd.clip import load_clip
    handle = load_clip(eng, args.clip_weights)
    model, _, pre = oc.create_model_and_transforms("ViT-L-14", pretrained=None)
    from facet_amd.pyiqa_scorer import load_checkpoint
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in load_checkpoint(args.clip_weights).items()}, strict=False)
    model.eval()
    from PIL import Image
    pil = [Image.fromarray(a) for a in images(args, 4, 640)]
    x = torch.stack([pre(im) for im in pil])
    with torch.no_grad():
        ref = model.encode_image(x).numpy()
    got = handle["model"].encode_image(torch.stack([handle["preprocess"](im) for im in pil]))
    report("image features", np.asarray(got), ref)
    report("preprocess tensor", torch.stack([handle["preprocess"](im) for im in pil]).numpy(), x.numpy())


def leg_faces(args, eng):
    print("SCRFD / 2d106det / ArcFace vs insightface on onnxruntime")
    if not args.buffalo_dir:
        print("  [skip] needs --buffalo-dir")
        return
    from facet_amd.face import FaceEngine
    files = {k: os.path.join(args.buffalo_dir, f) for k, f in (("det", "det_10g.onnx"), ("lmk", "2d106det.onnx"), ("rec", "w600k_r50.onnx"))}
    if not all(os.path.exists(p) for p in files.values()):
        print("  [skip] model files not found in", args.buffalo_dir)
        return
    fe = FaceEngine(engine=eng, models={k: open(p, "rb").read() for k, p in files.items()})
    bgr = images(args, 4, 1024)[..., ::-1].copy()
    mine = fe.get_batch(bgr)
    ia = have("insightface.app")
    if not ia:
        print("  engine results only:", [len(f) for f in mine], "faces per image")
        return
    app = ia.FaceAnalysis(name="buffalo_l", root=os.path.dirname(os.path.dirname(args.buffalo_dir.rstrip("/"))),
                          allowed_modules=["detection", "landmark_2d_106", "recognition"], providers=["CPUExecutionProvider"])
    app.prepare(ctx_id=-1, det_size=(640, 640))
    for i, img in enumerate(bgr):
        ref = app.get(img)
        print(f"  image {i}: engine {len(mine[i])} faces, insightface {len(ref)}")
        for a, b in zip(mine[i], ref):
            report("    bbox", a.bbox, b.bbox)
            report("    kps", a.kps, b.kps)
            report("    landmark_2d_106", a["landmark_2d_106"], b.landmark_2d_106)
            report("    embedding", a["embedding"], b.embedding)


def leg_cv(args, eng):
    print("pixel operators vs OpenCV")
    cv2 = have("cv2")
    if not cv2:
        return
    from facet_amd.composition import score_lines
    bgr = images(args, 3, 768)[..., ::-1].copy()
    st, gray, hsv = eng.image_stats(bgr, want_gray=True, want_hsv=True)
    lines, edges = eng.leading_lines(bgr, want_edges=True)
    for i, img in enumerate(bgr):
        g = cv2.cvtColor(img, cv2.COLOR_BGR2GRAY)
        print(f"  image {i}: gray equal {np.array_equal(gray[i], g)}, hsv equal {np.array_equal(hsv[i], cv2.cvtColor(img, cv2.COLOR_BGR2HSV))}")
        lap = cv2.Laplacian(g, cv2.CV_64F)
        report("    Laplacian sum / sum of squares", [st[i, 256], st[i, 257]], [lap.sum(), (lap * lap).sum()])
        e = cv2.Canny(cv2.GaussianBlur(g, (5, 5), 0), 50, 150)
        print(f"    Canny edge image equal {np.array_equal(edges[i], e)} ({int((edges[i] != e).sum())} pixels differ)")
        ref = cv2.HoughLinesP(e, 1, np.pi / 180, 80, minLineLength=int(min(img.shape[:2]) * 0.15), maxLineGap=20)
        ref = np.zeros((0, 4), np.int32) if ref is None else ref[:, 0, :]
        same = edges[i].tobytes() == e.tobytes() and np.array_equal(lines[i], ref)
        print(f"    HoughLinesP segments equal {same} (engine {len(lines[i])}, cv2 {len(ref)}; scores "
              f"{score_lines(lines[i], *img.shape[:2])['leading_lines_score']} vs {score_lines(ref, *img.shape[:2])['leading_lines_score']})")
        small = cv2.resize(img, (640, 640), interpolation=cv2.INTER_LINEAR)
        print(f"    cv2.resize INTER_LINEAR equal {np.array_equal(eng.cv_resize_linear(img[None], 640, 640)[0], small)}")


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--topiq-weights")
    ap.add_argument("--clip-weights")
    ap.add_argument("--buffalo-dir")
    ap.add_argument("--images", help="directory of photos (default: seeded synthetic images)")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()
    from facet_amd import Engine
    eng = Engine(args.device)                      # raises without a gfx950 device: there is no CPU path to pin
    for leg in (leg_topiq, leg_clip, leg_faces, leg_cv):
        try:
            leg(args, eng)
        except Exception as e:                      # noqa: BLE001 - one leg failing must not hide the others
            print(f"  [failed] {type(e).__name__}: {e}")
    eng.close()


if __name__ == "__main__":
    main()
