"""ORACLE -- test infrastructure only.  numpy-facing wrappers over postproc_oracle.c."""
import ctypes

import numpy as np

from . import load


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def bbox_iou(a, b):
    """a, b: (n,4) int32 xmin,ymin,xmax,ymax -> (n,) float64 (yolov3_detect.py:183-194)."""
    a = np.ascontiguousarray(a, np.int32); b = np.ascontiguousarray(b, np.int32)
    out = np.empty(len(a), np.float64)
    load().fvo_bbox_iou_batch(_p(a, ctypes.c_int), _p(b, ctypes.c_int), len(a), _p(out, ctypes.c_double))
    return out


def detect_postproc(head, image_size, conf_th, iou_th, num_cands):
    """head: (n, G, G, 6) float32 -> dict(boxes (n,K,4) i32, cell, obj, score, count).

    Restates FaceDetector.detect after model.predict (face_detection.py:900-949)."""
    head = np.ascontiguousarray(head, np.float32)
    n, g = head.shape[0], head.shape[1]
    k = max(int(num_cands), 1)
    boxes = np.full((n, k, 4), -1, np.int32); cell = np.full((n, k), -1, np.int32)
    obj = np.zeros((n, k), np.float32); score = np.zeros((n, k), np.float32)
    count = np.zeros(n, np.int32)
    load().fvo_detect_postproc_batch(_p(head, ctypes.c_float), n, g, int(image_size), float(conf_th),
                                     float(iou_th), int(num_cands), _p(boxes, ctypes.c_int),
                                     _p(cell, ctypes.c_int), _p(obj, ctypes.c_float),
                                     _p(score, ctypes.c_float), _p(count, ctypes.c_int))
    return dict(boxes=boxes, cell=cell, obj=obj, score=score, count=count)
