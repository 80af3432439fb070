"""TEST INFRASTRUCTURE ONLY - CPU restatement of the face path the reference runs through insightface + OpenCV.

Reference call sites: analyzers/face.py:30-38 (FaceAnalysis(name='buffalo_l', allowed_modules=[detection, landmark_2d_106,
recognition]).prepare(det_size=(640,640))), :99 (face_app.get(img_cv)), :101-234 (filtering and aggregation), :241-279 (EAR,
crop sharpness). The arithmetic itself lives in third-party packages absent offline (insightface>=0.7.0, onnxruntime,
opencv-python - requirements.txt:29-33), so every function restates the published algorithm [DEP-KNOWLEDGE] and names its source:
  insightface/model_zoo/scrfd.py (SCRFD.detect / forward / nms, distance2bbox, distance2kps)
  insightface/model_zoo/landmark.py (Landmark.get), arcface_onnx.py (ArcFaceONNX.get / get_feat)
  insightface/utils/face_align.py (estimate_norm, norm_crop, transform, trans_points2d), skimage _umeyama
  OpenCV imgproc: resize (INTER_LINEAR 8u), warpAffine (INTER_LINEAR, BORDER_CONSTANT), invertAffineTransform,
  cvtColor(BGR2GRAY), Laplacian(CV_64F)
Parity status: UNPINNED - no golden vectors exist in the reference for this path and none of the packages can be run here.
Networks are evaluated by oracle/onnx_ref.py on the same .onnx bytes the engine loads. Plain loops / float64 on purpose.
"""
import numpy as np

from . import onnx_ref

ARCFACE_DST = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366], [41.5493, 92.3655], [70.7299, 92.2041]],
                       dtype=np.float32)


# ---- OpenCV pieces ------------------------------------------------------------------------------------------------------
def _lin_tab(src, dst, clamp):
    scale = 1.0 / (float(dst) / float(src))
    ofs = np.zeros(dst, np.int64)
    coef = np.zeros((dst, 2), np.int64)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if clamp:
            if s < 0:
                f, s = np.float32(0), 0
            if s >= src - 1:
                f, s = np.float32(0), src - 1
        ofs[d] = s
        coef[d, 0] = int(np.rint(np.float32((np.float32(1) - f) * np.float32(2048))))
        coef[d, 1] = int(np.rint(np.float32(f * np.float32(2048))))
    return ofs, coef


def cv_resize_linear_u8(img, oh, ow):
    """cv2.resize(img, (ow, oh)) with the default INTER_LINEAR on a uint8 HxWx3 image."""
    h, w = img.shape[:2]
    if (h, w) == (oh, ow):
        return img.copy()
    a = img.astype(np.int64)
    if h == 2 * oh and w == 2 * ow:      # INTER_LINEAR at exactly 2x is routed to the 2x2 area average
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    xo, xa = _lin_tab(w, ow, True)
    yo, yb = _lin_tab(h, oh, False)
    x1 = np.minimum(xo + 1, w - 1)
    hor = a[:, xo] * xa[None, :, 0, None] + a[:, x1] * xa[None, :, 1, None]          # [h, ow, 3]
    y0 = np.clip(yo, 0, h - 1)
    y1 = np.clip(yo + 1, 0, h - 1)
    s0, s1 = hor[y0], hor[y1]
    v = (((yb[:, 0, None, None] * (s0 >> 4)) >> 16) + ((yb[:, 1, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def invert_affine(M):
    M = np.asarray(M, np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22, A12, A21 = M[1, 1] * D, M[0, 0] * D, -M[0, 1] * D, -M[1, 0] * D
    return np.array([[A11, A12, -A11 * M[0, 2] - A12 * M[1, 2]], [A21, A22, -A21 * M[0, 2] - A22 * M[1, 2]]], np.float64)


def _warp_table():
    tab = np.zeros((32, 32, 4), np.int64)
    for fy in range(32):
        for fx in range(32):
            ty = (np.float32(1) - np.float32(fy / 32.0), np.float32(fy / 32.0))
            tx = (np.float32(1) - np.float32(fx / 32.0), np.float32(fx / 32.0))
            t = [int(np.clip(np.rint(np.float32(ty[a] * tx[b] * np.float32(32768))), -32768, 32767)) for a in range(2) for b in range(2)]
            if sum(t) != 32768:
                t[3] -= sum(t) - 32768
            tab[fy, fx] = t
    return tab


_WTAB = None


def warp_affine_u8(img, M, size):
    """cv2.warpAffine(img, M, (size, size), borderValue=0.0) (INTER_LINEAR) on a uint8 HxWx3 image; M is the forward 2x3 matrix."""
    global _WTAB
    if _WTAB is None:
        _WTAB = _warp_table()
    h, w = img.shape[:2]
    Mi = invert_affine(M)
    out = np.zeros((size, size, 3), np.uint8)
    xs = np.arange(size)
    adelta = np.rint(Mi[0, 0] * xs * 1024.0).astype(np.int64)
    bdelta = np.rint(Mi[1, 0] * xs * 1024.0).astype(np.int64)
    a = img.astype(np.int64)
    for y in range(size):
        X0 = int(np.rint((Mi[0, 1] * y + Mi[0, 2]) * 1024.0)) + 16
        Y0 = int(np.rint((Mi[1, 1] * y + Mi[1, 2]) * 1024.0)) + 16
        X = (X0 + adelta) >> 5
        Y = (Y0 + bdelta) >> 5
        sx = np.clip(X >> 5, -32768, 32767)
        sy = np.clip(Y >> 5, -32768, 32767)
        wt = _WTAB[Y & 31, X & 31]                                  # [size, 4]
        acc = np.zeros((size, 3), np.int64)
        for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
            yy, xx = sy + dy, sx + dx
            ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            px = np.where(ok[:, None], a[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], 0)
            acc += px * wt[:, k, None]
        out[y] = np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    return out


def bgr2gray(img):
    b, g, r = (img[..., k].astype(np.int64) for k in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def laplacian_var(gray):
    """cv2.Laplacian(gray, cv2.CV_64F).var(): 4-neighbour kernel, BORDER_REFLECT_101."""
    g = gray.astype(np.float64)
    if g.size == 0:
        return 0.0
    p = np.pad(g, ((1, 1), (0, 0)), mode="reflect" if g.shape[0] > 1 else "edge")      # BORDER_REFLECT_101 per axis; a
    p = np.pad(p, ((0, 0), (1, 1)), mode="reflect" if g.shape[1] > 1 else "edge")      # length-1 axis repeats its only sample
    lap = p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] - 4.0 * g
    return float(lap.var())


# ---- insightface pieces ------------------------------------------------------------------------------------------------------
def nms(dets, thresh=0.4):
    x1, y1, x2, y2, scores = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        xx1 = np.maximum(x1[i], x1[order[1:]])
        yy1 = np.maximum(y1[i], y1[order[1:]])
        xx2 = np.minimum(x2[i], x2[order[1:]])
        yy2 = np.minimum(y2[i], y2[order[1:]])
        w = np.maximum(0.0, xx2 - xx1 + 1)
        h = np.maximum(0.0, yy2 - yy1 + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[order[1:]] - inter)
        order = order[np.where(ovr <= thresh)[0] + 1]
    return keep


def scrfd_detect(det_onnx, img, det_size=(640, 640), det_thresh=0.5, nms_thresh=0.4):
    """SCRFD.detect(img, input_size=det_size, max_num=0). img: BGR uint8. Returns (det [k,5] float32, kpss [k,5,2] float32)."""
    in_h, in_w = det_size
    im_ratio = float(img.shape[0]) / img.shape[1]
    model_ratio = float(in_h) / in_w
    if im_ratio > model_ratio:
        new_height = in_h
        new_width = int(new_height / im_ratio)
    else:
        new_width = in_w
        new_height = int(new_width * im_ratio)
    det_scale = float(new_height) / img.shape[0]
    resized = cv_resize_linear_u8(img, new_height, new_width)
    det_img = np.zeros((in_h, in_w, 3), np.uint8)
    det_img[:new_height, :new_width, :] = resized
    blob = ((det_img[:, :, ::-1].astype(np.float32) - np.float32(127.5)) * np.float32(1.0 / 128)).transpose(2, 0, 1)[None]
    outs = onnx_ref.run(det_onnx, blob)
    fmc, strides, na, use_kps = {6: (3, (8, 16, 32), 2, False), 9: (3, (8, 16, 32), 2, True), 10: (5, (8, 16, 32, 64, 128), 1, False),
                                 15: (5, (8, 16, 32, 64, 128), 1, True)}[len(outs)]
    scores_list, bboxes_list, kpss_list = [], [], []
    for idx, stride in enumerate(strides):
        scores = outs[idx]
        bbox_preds = outs[idx + fmc] * stride
        height, width = in_h // stride, in_w // stride
        centers = np.stack(np.mgrid[:height, :width][::-1], axis=-1).astype(np.float32)
        centers = (centers * stride).reshape((-1, 2))
        if na > 1:
            centers = np.stack([centers] * na, axis=1).reshape((-1, 2))
        pos = np.where(scores >= det_thresh)[0]
        bboxes = np.stack([centers[:, 0] - bbox_preds[:, 0], centers[:, 1] - bbox_preds[:, 1], centers[:, 0] + bbox_preds[:, 2],
                           centers[:, 1] + bbox_preds[:, 3]], axis=-1)
        scores_list.append(scores[pos])
        bboxes_list.append(bboxes[pos])
        if use_kps:
            kps_preds = outs[idx + 2 * fmc] * stride
            k = np.stack([centers[:, i % 2] + kps_preds[:, i] for i in range(kps_preds.shape[1])], axis=-1)
            kpss_list.append(k.reshape((k.shape[0], -1, 2))[pos])
    scores = np.vstack(scores_list)
    order = scores.ravel().argsort()[::-1]
    bboxes = np.vstack(bboxes_list) / det_scale
    pre_det = np.hstack((bboxes, scores)).astype(np.float32, copy=False)[order, :]
    keep = nms(pre_det, nms_thresh)
    det = pre_det[keep, :]
    kpss = None
    if use_kps:
        kpss = (np.vstack(kpss_list) / det_scale)[order, :, :][keep, :, :]
    return det, kpss


def umeyama(src, dst):
    """skimage.transform._geometric._umeyama(src, dst, estimate_scale=True) -> 3x3."""
    src = np.asarray(src, np.float64)
    dst = np.asarray(dst, np.float64)
    num, dim = src.shape
    src_mean, dst_mean = src.mean(axis=0), dst.mean(axis=0)
    src_demean, dst_demean = src - src_mean, dst - dst_mean
    A = dst_demean.T @ src_demean / num
    d = np.ones((dim,), np.float64)
    if np.linalg.det(A) < 0:
        d[dim - 1] = -1
    T = np.eye(dim + 1, dtype=np.float64)
    U, S, V = np.linalg.svd(A)
    rank = np.linalg.matrix_rank(A)
    if rank == 0:
        return np.nan * T
    if rank == dim - 1:
        if np.linalg.det(U) * np.linalg.det(V) > 0:
            T[:dim, :dim] = U @ V
        else:
            s = d[dim - 1]
            d[dim - 1] = -1
            T[:dim, :dim] = U @ np.diag(d) @ V
            d[dim - 1] = s
    else:
        T[:dim, :dim] = U @ np.diag(d) @ V
    scale = 1.0 / src_demean.var(axis=0).sum() * (S @ d)
    T[:dim, dim] = dst_mean - scale * (T[:dim, :dim] @ src_mean.T)
    T[:dim, :dim] *= scale
    return T


def estimate_norm(lmk, image_size=112):
    assert lmk.shape == (5, 2) and image_size % 112 == 0
    dst = ARCFACE_DST.astype(np.float64) * (float(image_size) / 112.0)
    return umeyama(lmk, dst)[0:2, :]


def landmark_transform(bbox, input_size=192):
    """The matrix Landmark.get builds with face_align.transform(img, center, input_size, scale, 0)."""
    bbox = [float(v) for v in bbox[:4]]
    w, h = bbox[2] - bbox[0], bbox[3] - bbox[1]
    cx, cy = (bbox[2] + bbox[0]) / 2, (bbox[3] + bbox[1]) / 2
    scale = input_size / (max(w, h) * 1.5)
    return np.array([[scale, 0.0, -cx * scale + input_size / 2], [0.0, scale, -cy * scale + input_size / 2]], np.float64)


def landmark_get(lmk_onnx, img, bbox, input_size=192, mean=0.0, std=1.0):
    M = landmark_transform(bbox, input_size)
    aimg = warp_affine_u8(img, M, input_size)
    blob = ((aimg[:, :, ::-1].astype(np.float32) - np.float32(mean)) * np.float32(1.0 / std)).transpose(2, 0, 1)[None]
    pred = onnx_ref.run(lmk_onnx, blob)[0][0].reshape((-1, 2)).astype(np.float32)
    pred[:, 0:2] += 1
    pred[:, 0:2] *= input_size // 2
    IM = invert_affine(M)
    out = np.zeros(pred.shape, np.float32)
    for i in range(pred.shape[0]):
        out[i] = (IM @ np.array([pred[i, 0], pred[i, 1], 1.0], np.float32))[0:2]
    return out


def arcface_get(rec_onnx, img, kps, mean=127.5, std=127.5):
    M = estimate_norm(np.asarray(kps, np.float32), 112)
    aimg = warp_affine_u8(img, M, 112)
    blob = ((aimg[:, :, ::-1].astype(np.float32) - np.float32(mean)) * np.float32(1.0 / std)).transpose(2, 0, 1)[None]
    return onnx_ref.run(rec_onnx, blob)[0].flatten()


def face_analysis_get(models, img, det_size=(640, 640), det_thresh=0.5):
    """FaceAnalysis.get(img): list of dicts(bbox, kps, det_score, landmark_2d_106, embedding). models: dict det/lmk/rec ->
    (onnx bytes, mean, std)."""
    det, kpss = scrfd_detect(models["det"][0], img, det_size, det_thresh)
    faces = []
    for i in range(det.shape[0]):
        f = {"bbox": det[i, 0:4], "det_score": det[i, 4], "kps": kpss[i] if kpss is not None else None}
        if "lmk" in models:
            f["landmark_2d_106"] = landmark_get(models["lmk"][0], img, f["bbox"], 192, models["lmk"][1], models["lmk"][2])
        if "rec" in models and f["kps"] is not None:
            f["embedding"] = arcface_get(models["rec"][0], img, f["kps"], models["rec"][1], models["rec"][2])
        faces.append(f)
    return faces


# ---- reference analyzers/face.py:101-234 ---------------------------------------------------------------------------------
LEFT_EYE = [35, 39, 37, 38, 41, 40]
RIGHT_EYE = [89, 93, 91, 92, 95, 94]


def ear(landmarks, idx):
    v1 = np.linalg.norm(landmarks[idx[2]] - landmarks[idx[4]])
    v2 = np.linalg.norm(landmarks[idx[3]] - landmarks[idx[5]])
    hh = np.linalg.norm(landmarks[idx[0]] - landmarks[idx[1]])
    return (v1 + v2) / (2.0 * hh) if hh > 0 else 0.3


def analyze_faces(all_faces, img, min_confidence=0.7, min_face_size=30, blink_ear_threshold=0.21, min_faces_for_group=4):
    """The dict reference FaceAnalyzer.analyze_faces builds from FaceAnalysis.get's faces (thumbnails excluded)."""
    faces, max_conf = [], 0
    for f in all_faces:
        conf = float(f["det_score"])
        max_conf = max(max_conf, conf)
        if conf < min_confidence:
            continue
        bb = f["bbox"].astype(int)
        if bb[2] - bb[0] < min_face_size or bb[3] - bb[1] < min_face_size:
            continue
        faces.append(f)
    zeros = {'face_count': 0, 'face_quality': 0, 'eye_sharpness': 0, 'is_blink': 0, 'face_area': 0, 'bbox': None, 'face_sharpness': 0,
             'raw_eye_sharpness': 0, 'is_group_portrait': 0, 'max_face_confidence': max_conf, 'face_details': []}
    if not faces:
        return zeros
    h, w = img.shape[:2]
    quals, eyes, raw_eyes, sharp = [], [], [], []
    any_blink, area = False, 0
    mnx, mny, mxx, mxy = w, h, 0, 0
    for f in faces:
        bb = f["bbox"].astype(int)
        mnx, mny, mxx, mxy = min(mnx, bb[0]), min(mny, bb[1]), max(mxx, bb[2]), max(mxy, bb[3])
        quals.append(float(f["det_score"] * 10))
        eye_score = 0
        lm = f.get("landmark_2d_106")
        if lm is not None:
            l_eye, r_eye = lm[38], lm[92]
            off = int(np.linalg.norm(l_eye - r_eye) * 0.15)
            ev = []
            for ex, ey in (l_eye, r_eye):
                ex1, ex2, ey1, ey2 = int(ex - off), int(ex + off), int(ey - off), int(ey + off)
                roi = img[max(0, ey1):min(h, ey2), max(0, ex1):min(w, ex2)]
                if roi.size > 0:
                    g = bgr2gray(roi)
                    ev.append(laplacian_var(g) / (np.mean(g) + 1))
            eye_score = max(ev) if ev else 0
        eyes.append(min(10.0, eye_score / 2.0))
        raw_eyes.append(eye_score)
        y1, y2, x1, x2 = max(0, bb[1]), min(h, bb[3]), max(0, bb[0]), min(w, bb[2])
        crop = img[y1:y2, x1:x2]
        sharp.append(0 if crop.size == 0 else laplacian_var(bgr2gray(crop)))
        if lm is not None and (ear(lm, LEFT_EYE) + ear(lm, RIGHT_EYE)) / 2.0 < blink_ear_threshold:
            any_blink = True
        area += (bb[2] - bb[0]) * (bb[3] - bb[1])
    return {'face_count': len(faces), 'face_quality': round(0.7 * min(quals) + 0.3 * (sum(quals) / len(quals)), 2),
            'eye_sharpness': round(sum(eyes) / len(eyes), 2), 'raw_eye_sharpness': sum(raw_eyes) / len(raw_eyes),
            'face_sharpness': sum(sharp) / len(sharp), 'is_blink': 1 if any_blink else 0, 'face_area': area,
            'bbox': np.array([mnx, mny, mxx, mxy]), 'is_group_portrait': 1 if len(faces) >= min_faces_for_group else 0,
            'max_face_confidence': max_conf,
            'face_details': [{'index': i, 'bbox': f["bbox"].astype(int).tolist(), 'confidence': float(f["det_score"])} for i, f in enumerate(faces)]}
