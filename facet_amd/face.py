"""Drop-in for reference analyzers/face.py `FaceAnalyzer`, with InsightFace's FaceAnalysis pipeline served by libfacet_engine.so.

Reference behaviour mirrored (analyzers/face.py): constructor arguments and attributes (:15-29), `.available` (False with the
"InsightFace not available: ..." message when the models cannot be loaded, :39-40), `analyze_faces(BGR ndarray) -> dict` with the
same keys, filtering (confidence, min size :101-122), aggregation (:134-234), EAR helpers (:236-270) and crop sharpness (:272-279).

What runs where:
  GPU (engine): SCRFD preprocessing (cv2.resize + canvas + blob), detector graph, threshold/decode; for ALL faces of a batch
      in one call each: similarity-warped 112x112 crops + ArcFace graph, 192x192 crops + 2d106 landmark graph
      (fe_face_detect / fe_face_crops_run, include/facet_engine.h). The reference runs these per image and per face.
  host (here): sort + NMS over the few candidates, 5-point similarity estimate, landmark back-projection, the reference's own
      post-processing (gray / Laplacian variance on small ROIs, EAR, aggregation) and JPEG thumbnails.
The three networks are the reference's own files: <root>/models/buffalo_l/{det_10g,2d106det,w600k_r50}.onnx (insightface's
layout, root='~/.insightface' at face.py:34), parsed by the engine's ONNX runtime; `models=` passes bytes directly (tests use
standins.synthetic_onnx). insightface internals follow the published package [DEP-KNOWLEDGE]; cv2 is not importable here, so
gray/Laplacian are restated in numpy and thumbnails are encoded with Pillow (not bit-identical to cv2.imencode).
"""
import io
import os

import numpy as np

from ._lib import Engine, EngineError, FE_GRAPH_FACE_DET, FE_GRAPH_FACE_LMK, FE_GRAPH_FACE_REC

ARCFACE_DST = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366], [41.5493, 92.3655], [70.7299, 92.2041]],
                       dtype=np.float32)
BUFFALO_L = {"det": "det_10g.onnx", "lmk": "2d106det.onnx", "rec": "w600k_r50.onnx"}


class Face(dict):
    """insightface.app.common.Face: a dict whose keys read as attributes (missing -> None)."""

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return self.get(name)

    def __setattr__(self, name, value):
        self[name] = value


def similarity_from_5pts(src, dst):
    """Least-squares similarity (rotation, uniform scale, translation) src -> dst, the transform skimage's
    SimilarityTransform.estimate (Umeyama) yields for 2-D points. Closed form: the best proper rotation maximises
    trace(R^T A), giving angle atan2(A10 - A01, A00 + A11) and scale |.| / var(src)."""
    src = np.asarray(src, np.float64)
    dst = np.asarray(dst, np.float64)
    sm, dm = src.mean(axis=0), dst.mean(axis=0)
    sd, dd = src - sm, dst - dm
    A = dd.T @ sd / src.shape[0]
    p, q = A[0, 0] + A[1, 1], A[1, 0] - A[0, 1]
    r = np.hypot(p, q)
    if r == 0:
        return np.full((2, 3), np.nan)
    c, s = p / r, q / r
    scale = r / sd.var(axis=0).sum()
    R = np.array([[c, -s], [s, c]]) * scale
    t = dm - R @ sm
    return np.array([[R[0, 0], R[0, 1], t[0]], [R[1, 0], R[1, 1], t[1]]], np.float64)


def invert_affine(M):
    """cv2.invertAffineTransform for [m,2,3] matrices."""
    M = np.asarray(M, np.float64)
    D = M[:, 0, 0] * M[:, 1, 1] - M[:, 0, 1] * M[:, 1, 0]
    D = np.where(D != 0, 1.0 / np.where(D != 0, D, 1.0), 0.0)
    A11, A22, A12, A21 = M[:, 1, 1] * D, M[:, 0, 0] * D, -M[:, 0, 1] * D, -M[:, 1, 0] * D
    out = np.empty_like(M)
    out[:, 0, 0], out[:, 0, 1], out[:, 0, 2] = A11, A12, -A11 * M[:, 0, 2] - A12 * M[:, 1, 2]
    out[:, 1, 0], out[:, 1, 1], out[:, 1, 2] = A21, A22, -A21 * M[:, 0, 2] - A22 * M[:, 1, 2]
    return out


def nms(dets, thresh=0.4):
    """insightface SCRFD.nms on score-sorted rows [x1,y1,x2,y2,score]; returns kept row indices."""
    if dets.shape[0] == 0:
        return []
    x1, y1, x2, y2 = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = dets[:, 4].argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        rest = order[1:]
        w = np.maximum(0.0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
        h = np.maximum(0.0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        order = rest[ovr <= thresh]
    return keep


def bgr2gray(img):
    """cv2.cvtColor(img, COLOR_BGR2GRAY) for uint8: 15-bit fixed point (R 9798, G 19235, B 3735; OpenCV 4.x)."""
    a = img.astype(np.int32)
    return ((a[..., 0] * 3735 + a[..., 1] * 19235 + a[..., 2] * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def laplacian_var(gray):
    """cv2.Laplacian(gray, cv2.CV_64F).var(): aperture 1 (4-neighbour), BORDER_REFLECT_101."""
    if gray.size == 0:
        return 0.0
    g = gray.astype(np.float64)
    p = np.pad(g, ((1, 1), (0, 0)), mode="reflect" if g.shape[0] > 1 else "edge")      # BORDER_REFLECT_101 per axis; a
    p = np.pad(p, ((0, 0), (1, 1)), mode="reflect" if g.shape[1] > 1 else "edge")      # length-1 axis repeats its only sample
    lap = p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] - 4.0 * g
    return float(lap.var())


class _GraphsHandle:
    """What ModelManager pokes at when it parks a model in RAM: .cpu() frees the device graphs, .to(dev) loads them again."""

    def __init__(self, fe):
        self._fe = fe

    def cpu(self):
        self._fe.unload()
        return self

    def to(self, device):
        if str(device) == 'cpu':
            self._fe.unload()
        else:
            self._fe.load()
        return self

    def eval(self):
        return self


class FaceEngine:
    """insightface.app.FaceAnalysis for batches: detection -> landmark_2d_106 -> recognition on the engine."""

    def __init__(self, engine, models, det_size=(640, 640), det_thresh=0.5, nms_thresh=0.4, max_candidates=1024, max_faces=64):
        self.engine = engine
        self.max_faces = max_faces
        self._models = dict(models)
        self.model = _GraphsHandle(self)
        self.det_size, self.det_thresh, self.nms_thresh, self.max_candidates = det_size, det_thresh, nms_thresh, max_candidates
        if "det" not in models:
            raise EngineError("face models: a detection model is required")     # FaceAnalysis asserts 'detection' in models
        self.has = {k: k in models for k in ("det", "lmk", "rec")}
        for key, slot in (("det", FE_GRAPH_FACE_DET), ("lmk", FE_GRAPH_FACE_LMK), ("rec", FE_GRAPH_FACE_REC)):
            if key in models:
                engine.graph_load(slot, models[key])
        # input normalisation as insightface picks it: Sub/Mul nodes at the head of the graph -> the model normalises itself
        self.norm = {}
        for key, slot, default_std in (("lmk", FE_GRAPH_FACE_LMK, 128.0), ("rec", FE_GRAPH_FACE_REC, 127.5)):
            if self.has[key]:
                info = engine.graph_info(slot)
                self.norm[key] = (0.0, 1.0) if (info["has_sub"] and info["has_mul"]) else (127.5, default_std)
                self.norm[key + "_size"] = int(info["input_dims"][2]) if info["input_dims"][2] > 0 else (192 if key == "lmk" else 112)

    def load(self):
        for key, slot in (("det", FE_GRAPH_FACE_DET), ("lmk", FE_GRAPH_FACE_LMK), ("rec", FE_GRAPH_FACE_REC)):
            if key in self._models and not self.engine.graph_loaded(slot):
                self.engine.graph_load(slot, self._models[key])

    def unload(self):
        for slot in (FE_GRAPH_FACE_DET, FE_GRAPH_FACE_LMK, FE_GRAPH_FACE_REC):
            if self.engine.graph_loaded(slot):
                self.engine.graph_unload(slot)

    def detect(self, images):
        """images: BGR uint8 [n,h,w,3] -> per image (det [k,5] float32 sorted by score, kps [k,5,2] float32)."""
        cand, counts, _ = self.engine.face_detect(images, self.det_size, self.det_thresh, self.max_candidates)
        out = []
        for i in range(cand.shape[0]):
            k = int(min(counts[i], self.max_candidates))
            c = cand[i, :k]
            # SCRFD.detect sorts all levels by score before NMS; ties broken by (level, position) to be deterministic
            order = np.lexsort((c[:, 2], c[:, 1], c[:, 15], -c[:, 0]))
            c = c[order]
            pre = np.concatenate([c[:, 1:5], c[:, 0:1]], axis=1).astype(np.float32)
            keep = nms(pre, self.nms_thresh)
            out.append((pre[keep], c[keep, 5:15].reshape(-1, 5, 2).astype(np.float32)))
        return out

    def get_batch(self, images):
        """FaceAnalysis.get for every image of a same-sized BGR batch; returns list (per image) of list[Face].
        One engine call (fe_face_analyze): detection, NMS, landmarks and embeddings for the whole batch. Images with more than
        `max_faces` detections keep the best-scoring max_faces (insightface keeps all; raise max_faces if that matters)."""
        if not isinstance(images, tuple):       # (device_ptr, n, h, w) batches are used in place
            images = np.ascontiguousarray(images, dtype=np.uint8)
        rec, counts, mask = self.engine.face_analyze(images, self.det_size, self.det_thresh, self.nms_thresh, self.max_faces)
        out = []
        for i in range(rec.shape[0]):
            faces = []
            for f in range(min(int(counts[i]), self.max_faces)):
                r = rec[i, f]
                face = Face(bbox=r[0:4].copy(), det_score=r[4], kps=r[5:15].reshape(5, 2).copy())
                if mask & 2:
                    face['landmark_2d_106'] = r[15:227].reshape(106, 2).copy()
                if mask & 4:
                    face['embedding'] = r[227:739].copy()
                faces.append(face)
            out.append(faces)
        return out

    def get_batch_host(self, images):
        """The same pipeline with the glue (sort, NMS, crop matrices, back-projection) in numpy on top of fe_face_detect /
        fe_face_crops_run. Kept as the readable statement of what fe_face_analyze does natively; tests cross-check the two."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        n, h, w, _ = images.shape
        e = self.engine
        d = e.dev_alloc(images.nbytes)
        try:
            e.h2d(d, images)
            dev = (d, n, h, w)
            dets = self.detect(dev)
            img_idx = np.concatenate([np.full(det.shape[0], i, np.int32) for i, (det, _) in enumerate(dets)]) if dets else np.zeros(0, np.int32)
            boxes = np.concatenate([det for det, _ in dets]) if dets else np.zeros((0, 5), np.float32)
            kpss = np.concatenate([k for _, k in dets]) if dets else np.zeros((0, 5, 2), np.float32)
            m = boxes.shape[0]
            lmk = emb = None
            if m and self.has["lmk"]:
                S = self.norm["lmk_size"]
                b = boxes[:, :4].astype(np.float64)
                bw, bh = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
                cx, cy = (b[:, 2] + b[:, 0]) / 2, (b[:, 3] + b[:, 1]) / 2
                sc = S / (np.maximum(bw, bh) * 1.5)
                M = np.zeros((m, 2, 3), np.float64)
                M[:, 0, 0] = M[:, 1, 1] = sc
                M[:, 0, 2] = -cx * sc + S / 2
                M[:, 1, 2] = -cy * sc + S / 2
                mean, std = self.norm["lmk"]
                pred, _ = e.face_crops_run(FE_GRAPH_FACE_LMK, dev, img_idx, M, S, mean, 1.0 / std, True, out_dim=212)
                pred = pred.reshape(m, -1, 2)
                pred = (pred + np.float32(1)) * np.float32(S // 2)
                IM = invert_affine(M)
                hom = np.concatenate([pred, np.ones((m, pred.shape[1], 1), np.float32)], axis=2).astype(np.float64)
                lmk = np.einsum('mij,mkj->mki', IM, hom).astype(np.float32)
            if m and self.has["rec"]:
                S = self.norm["rec_size"]
                dst = ARCFACE_DST.astype(np.float64) * (float(S) / 112.0)
                M = np.stack([similarity_from_5pts(kpss[f], dst) for f in range(m)])
                mean, std = self.norm["rec"]
                emb, _ = e.face_crops_run(FE_GRAPH_FACE_REC, dev, img_idx, M, S, mean, 1.0 / std, True, out_dim=512)
        finally:
            e.dev_free(d)
        faces = [[] for _ in range(n)]
        for f in range(m):
            face = Face(bbox=boxes[f, :4], kps=kpss[f], det_score=boxes[f, 4])
            if lmk is not None:
                face['landmark_2d_106'] = lmk[f]
            if emb is not None:
                face['embedding'] = emb[f]
            faces[int(img_idx[f])].append(face)
        return faces

    def get(self, img):
        return self.get_batch(np.asarray(img)[None])[0]


def _load_buffalo_l(root):
    d = os.path.join(os.path.expanduser(root), 'models', 'buffalo_l')
    models = {}
    for key, fn in BUFFALO_L.items():
        p = os.path.join(d, fn)
        if os.path.exists(p):
            with open(p, 'rb') as f:
                models[key] = f.read()
    if "det" not in models:
        raise FileNotFoundError(f"{os.path.join(d, BUFFALO_L['det'])} not found (the engine does not download models)")
    return models


class FaceAnalyzer:
    # 106-point landmark indices: [outer, inner, upper, upper2, lower, lower2] (reference face.py:238-239)
    LEFT_EYE_INDICES = [35, 39, 37, 38, 41, 40]
    RIGHT_EYE_INDICES = [89, 93, 91, 92, 95, 94]

    def __init__(self, device='cuda', min_confidence=0.7, min_face_size=30, thumbnail_size=128, thumbnail_quality=85,
                 blink_ear_threshold=0.21, min_faces_for_group=4, engine=None, models=None, root='~/.insightface'):
        self.available = False
        self.min_confidence = min_confidence
        self.min_face_size = min_face_size
        self.thumbnail_size = thumbnail_size
        self.thumbnail_quality = thumbnail_quality
        self.blink_ear_threshold = blink_ear_threshold
        self.min_faces_for_group = min_faces_for_group
        self.face_app = None
        try:
            if models is None:
                models = _load_buffalo_l(root)
            if engine is None:
                idx = int(str(device).split(':')[1]) if ':' in str(device) else 0
                engine = Engine(idx)
            self.face_app = FaceEngine(engine, models, det_size=(640, 640))
            self.available = True
        except Exception as e:   # same contract as the reference: report and stay unavailable (face.py:39-40)
            print(f"InsightFace not available: {e}")

    def _crop_face_thumbnail(self, img_cv, bbox, padding=0.3):
        """JPEG bytes of the face box grown by `padding` of its size on every side (clipped to the image) and scaled so its longer edge
        is `thumbnail_size` - the arithmetic of the reference's helper (analyzers/face.py:52-82: int() truncation of the box, of the
        padding and of the scaled size), with Pillow doing the resampling and encoding (cv2 is not available: pixels are close to,
        not identical with, the reference's INTER_AREA + cv2.imencode bytes)."""
        try:
            from PIL import Image
            left, top, right, bottom = (int(v) for v in bbox)
            grow_x, grow_y = int((right - left) * padding), int((bottom - top) * padding)
            rows = slice(max(0, top - grow_y), min(img_cv.shape[0], bottom + grow_y))
            cols = slice(max(0, left - grow_x), min(img_cv.shape[1], right + grow_x))
            crop = img_cv[rows, cols]
            if crop.size == 0:
                return None
            factor = self.thumbnail_size / max(crop.shape[0], crop.shape[1])
            size = (int(crop.shape[1] * factor), int(crop.shape[0] * factor))            # PIL takes (width, height)
            thumb = Image.fromarray(np.ascontiguousarray(crop[:, :, ::-1])).resize(size, Image.BOX)
            out = io.BytesIO()
            thumb.save(out, format='JPEG', quality=int(self.thumbnail_quality))
            return out.getvalue()
        except Exception:
            return None

    @staticmethod
    def _zeros(max_conf=0):
        return {'face_count': 0, 'face_quality': 0, 'eye_sharpness': 0, 'is_blink': 0, 'face_area': 0, 'bbox': None,
                'face_sharpness': 0, 'raw_eye_sharpness': 0, 'is_group_portrait': 0, 'max_face_confidence': max_conf,
                'face_details': []}

    def analyze_faces(self, img_cv):
        if not self.available or img_cv is None:
            return self._zeros()
        return self._post(self.face_app.get(img_cv), img_cv)

    def analyze_faces_batch(self, images, resident=None):
        """Same-sized BGR images -> list of analyze_faces dicts, with every network run once per batch. resident: optional
        (device_ptr, n, h, w) of the same BGR batch already in device memory - then `images` (any per-image array-likes, e.g.
        reversed-channel views of an RGB batch) are only read for the thumbnails of the faces found."""
        if not self.available or images is None or len(images) == 0:
            return [self._zeros() for _ in (images if images is not None else [])]
        e = self.face_app.engine
        if resident is None:
            arr = np.ascontiguousarray(np.stack([np.asarray(im) for im in images]), dtype=np.uint8)
            d = e.dev_alloc(arr.nbytes)
        else:
            arr, d = images, None
            assert resident[1] == len(images)
        try:
            if resident is None:
                e.h2d(d, arr)
                dev = (d, arr.shape[0], arr.shape[1], arr.shape[2])
            else:
                dev = resident
            per_image = self.face_app.get_batch(dev)
            # first pass only records which ROIs the reference logic looks at; one engine call scans them all
            wanted = []
            for i, faces in enumerate(per_image):
                self._post(faces, arr[i], lambda x1, y1, x2, y2, i=i: (wanted.append((i, x1, y1, x2, y2)), (0.0, 0.0))[1], thumbnails=False)
            table = {}
            if wanted:
                st = e.roi_laplacian(dev, [r[0] for r in wanted], [r[1:] for r in wanted])
                for r, (ls, lss, gs, cnt) in zip(wanted, st):
                    mean = ls / cnt
                    table[r] = (lss / cnt - mean * mean, gs / cnt)
        finally:
            if d is not None:
                e.dev_free(d)
        return [self._post(faces, arr[i], lambda x1, y1, x2, y2, i=i: table[(i, x1, y1, x2, y2)]) for i, faces in enumerate(per_image)]

    @staticmethod
    def _roi_numpy(img_cv):
        def fn(x1, y1, x2, y2):
            g = bgr2gray(img_cv[y1:y2, x1:x2])
            return laplacian_var(g), float(np.mean(g))
        return fn

    def _post(self, all_faces, img_cv, roi_stats=None, thumbnails=True):
        """reference analyze_faces :101-234. roi_stats(x1,y1,x2,y2) -> (Laplacian variance, mean gray) of a non-empty clipped ROI;
        default: numpy on the host image (single-image path), batch path: fe_roi_laplacian results."""
        if roi_stats is None:
            roi_stats = self._roi_numpy(img_cv)
        faces = []
        max_confidence = 0
        for face in all_faces:
            confidence = float(face.det_score)
            max_confidence = max(max_confidence, confidence)
            if confidence < self.min_confidence:
                continue
            bbox = face.bbox.astype(int)
            if bbox[2] - bbox[0] < self.min_face_size or bbox[3] - bbox[1] < self.min_face_size:
                continue
            faces.append(face)
        if not faces:
            return self._zeros(max_confidence)
        h, w = img_cv.shape[:2]
        all_qualities, all_eye_scores, all_raw_eye_scores, all_face_sharpness = [], [], [], []
        any_blink = False
        total_face_area = 0
        min_x, min_y, max_x, max_y = w, h, 0, 0
        for face in faces:
            bbox = face.bbox.astype(int)
            min_x, min_y = min(min_x, bbox[0]), min(min_y, bbox[1])
            max_x, max_y = max(max_x, bbox[2]), max(max_y, bbox[3])
            all_qualities.append(float(face.det_score * 10))
            eye_score = 0
            if face.landmark_2d_106 is not None:
                l_eye, r_eye = face.landmark_2d_106[38], face.landmark_2d_106[92]
                offset = int(np.linalg.norm(l_eye - r_eye) * 0.15)
                eye_vars = []
                for ex, ey in [l_eye, r_eye]:
                    ex1, ex2 = int(ex - offset), int(ex + offset)
                    ey1, ey2 = int(ey - offset), int(ey + offset)
                    # the reference slices img_cv[max(0,ey1):min(h,ey2), max(0,ex1):min(w,ex2)]: a negative upper bound counts
                    # from the far edge in numpy; slice.indices reproduces exactly the region numpy would cut
                    ry1, ry2, _ = slice(max(0, ey1), min(h, ey2)).indices(h)
                    rx1, rx2, _ = slice(max(0, ex1), min(w, ex2)).indices(w)
                    if rx2 > rx1 and ry2 > ry1:          # eye_roi.size > 0
                        var, mean_gray = roi_stats(rx1, ry1, rx2, ry2)
                        eye_vars.append(var / (mean_gray + 1))
                eye_score = max(eye_vars) if eye_vars else 0
            all_eye_scores.append(min(10.0, eye_score / 2.0))
            all_raw_eye_scores.append(eye_score)
            all_face_sharpness.append(self._get_crop_sharpness(img_cv, bbox, roi_stats))
            if self.is_blinking(face):
                any_blink = True
            total_face_area += (bbox[2] - bbox[0]) * (bbox[3] - bbox[1])
        min_quality = min(all_qualities)
        avg_quality = sum(all_qualities) / len(all_qualities)
        face_details = []
        for idx, face in enumerate(faces):
            bbox = face.bbox.astype(int)
            face_details.append({
                'index': idx,
                'bbox': bbox.tolist(),
                'confidence': float(face.det_score),
                'embedding': face.embedding.astype(np.float32).tobytes() if face.embedding is not None else None,
                'landmark_2d_106': face.landmark_2d_106.astype(np.float32).tobytes() if face.landmark_2d_106 is not None else None,
                'thumbnail': self._crop_face_thumbnail(img_cv, bbox) if thumbnails else None,
            })
        return {
            'face_obj': faces[0],
            'face_count': len(faces),
            'face_quality': round(0.7 * min_quality + 0.3 * avg_quality, 2),
            'eye_sharpness': round(sum(all_eye_scores) / len(all_eye_scores), 2),
            'raw_eye_sharpness': sum(all_raw_eye_scores) / len(all_raw_eye_scores),
            'face_sharpness': sum(all_face_sharpness) / len(all_face_sharpness),
            'is_blink': 1 if any_blink else 0,
            'face_area': total_face_area,
            'bbox': np.array([min_x, min_y, max_x, max_y]),
            'is_group_portrait': 1 if len(faces) >= self.min_faces_for_group else 0,
            'max_face_confidence': max_confidence,
            'face_details': face_details,
        }

    @staticmethod
    def calculate_ear(landmarks, eye_indices):
        lm = np.asarray(landmarks)
        v1 = np.linalg.norm(lm[eye_indices[2]] - lm[eye_indices[4]])
        v2 = np.linalg.norm(lm[eye_indices[3]] - lm[eye_indices[5]])
        h = np.linalg.norm(lm[eye_indices[0]] - lm[eye_indices[1]])
        return (v1 + v2) / (2.0 * h) if h > 0 else 0.3

    @staticmethod
    def compute_avg_ear(landmarks):
        return (FaceAnalyzer.calculate_ear(landmarks, FaceAnalyzer.LEFT_EYE_INDICES) +
                FaceAnalyzer.calculate_ear(landmarks, FaceAnalyzer.RIGHT_EYE_INDICES)) / 2.0

    def is_blinking(self, face):
        lm = face.get('landmark_2d_106') if isinstance(face, dict) else getattr(face, 'landmark_2d_106', None)
        if lm is None:
            return False
        return self.compute_avg_ear(lm) < self.blink_ear_threshold

    def _get_crop_sharpness(self, img, bbox, roi_stats=None):
        h, w = img.shape[:2]
        y1, y2, _ = slice(int(max(0, bbox[1])), int(min(h, bbox[3]))).indices(h)      # numpy slice semantics, as above
        x1, x2, _ = slice(int(max(0, bbox[0])), int(min(w, bbox[2]))).indices(w)
        if y2 <= y1 or x2 <= x1:                 # crop.size == 0
            return 0
        return (roi_stats or self._roi_numpy(img))(int(x1), int(y1), int(x2), int(y2))[0]
