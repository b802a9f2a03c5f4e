"""ORACLE -- test infrastructure only.

CPU restatements of the reference's hot-path algorithms, used as the checker by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
face_vijnana_yolov3_amd/ imports this package.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, 'libfv_oracle.so')
    src = os.path.join(_HERE, 'postproc_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'libfv_oracle.so'])
    return so


def load():
    """ctypes handle on the C oracle (built on first use)."""
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build())
        c_int_p = ctypes.POINTER(ctypes.c_int)
        c_f_p = ctypes.POINTER(ctypes.c_float)
        lib.fvo_sigmoid_f32.restype = ctypes.c_float
        lib.fvo_sigmoid_f32.argtypes = [ctypes.c_float]
        lib.fvo_bbox_iou.restype = ctypes.c_double
        lib.fvo_bbox_iou.argtypes = [c_int_p, c_int_p]
        lib.fvo_bbox_iou_batch.restype = None
        lib.fvo_bbox_iou_batch.argtypes = [c_int_p, c_int_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        lib.fvo_detect_postproc_batch.restype = None
        lib.fvo_detect_postproc_batch.argtypes = [c_f_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                                  c_int_p, c_int_p, c_f_p, c_f_p, c_int_p]
        _LIB = lib
    return _LIB
