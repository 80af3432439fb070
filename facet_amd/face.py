"""Interface mirror of reference analyzers/face.py `FaceAnalyzer` — WITHOUT a detector.

InsightFace buffalo_l (SCRFD-10GF, 2d106det, ArcFace-R50) is served to the reference by onnxruntime from ONNX files
that are not available offline, so the engine has no face models yet.
This mirror keeps the call surface the orchestrator depends on, with the reference's own "unavailable" behaviour:
  FaceAnalyzer(device, min_confidence, min_face_size, thumbnail_size, thumbnail_quality, blink_ear_threshold,
               min_faces_for_group); .available; .analyze_faces(BGR ndarray) -> dict; static compute_avg_ear(landmarks).
`available` is False, so analyze_faces returns the zeros dict exactly as the reference does when InsightFace fails
to import (face.py:90-97); the EAR helpers (pure arithmetic, face.py:236-270) are complete because
processing/scorer.py:1405 calls FaceAnalyzer.compute_avg_ear on stored landmarks.
"""
import numpy as np


class FaceAnalyzer:
    # 106-point landmark indices: [outer, inner, upper, upper2, lower, lower2] (reference face.py:238-239)
    LEFT_EYE_INDICES = [35, 39, 37, 38, 41, 40]
    RIGHT_EYE_INDICES = [89, 93, 91, 92, 95, 94]

    def __init__(self, device='cuda', min_confidence=0.7, min_face_size=30, thumbnail_size=128, thumbnail_quality=85,
                 blink_ear_threshold=0.21, min_faces_for_group=4):
        self.available = False
        self.min_confidence = min_confidence
        self.min_face_size = min_face_size
        self.thumbnail_size = thumbnail_size
        self.thumbnail_quality = thumbnail_quality
        self.blink_ear_threshold = blink_ear_threshold
        self.min_faces_for_group = min_faces_for_group
        print("InsightFace not available: the MI355X engine has no SCRFD/ArcFace graphs yet")

    def analyze_faces(self, img_cv):
        # reference face.py:90-97 (the only branch reachable while available is False)
        return {
            'face_count': 0, 'face_quality': 0, 'eye_sharpness': 0,
            'is_blink': 0, 'face_area': 0, 'bbox': None,
            'face_sharpness': 0, 'raw_eye_sharpness': 0,
            'is_group_portrait': 0, 'max_face_confidence': 0,
            'face_details': []
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
        if not hasattr(face, 'landmark_2d_106'):
            return False
        return self.compute_avg_ear(face.landmark_2d_106) < self.blink_ear_threshold
