"""ctypes binding of include/fv_hotpath.h (the same stub INTEGRATION.md shows)."""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, 'libfv_hotpath.so')
_lib = None

c_void_p, c_int, c_double, c_char_p = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_char_p


class FvError(RuntimeError):
    pass


def lib():
    """Load libfv_hotpath.so; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FvError('libfv_hotpath.so not built: run `python -m face_vijnana_yolov3_amd.build` '
                          '(there is no CPU fallback for the hot path)')
        L = ctypes.CDLL(LIB_PATH)
        L.fv_abi_version.restype = c_int
        L.fv_create.restype = c_int
        L.fv_create.argtypes = [c_int, c_void_p, ctypes.POINTER(c_void_p)]
        L.fv_destroy.restype = None
        L.fv_destroy.argtypes = [c_void_p]
        L.fv_last_error.restype = c_char_p
        L.fv_last_error.argtypes = [c_void_p]
        L.fv_set_stream.restype = c_int
        L.fv_set_stream.argtypes = [c_void_p, c_void_p]
        L.fv_decode_nms.restype = c_int
        L.fv_decode_nms.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_double, c_int,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        _lib = L
    return _lib


class Context:
    """One fv_ctx per GPU / rank, bound to a HIP stream (default: torch's current stream)."""

    def __init__(self, device=0, stream=None):
        import torch
        if not torch.cuda.is_available():
            raise FvError('no MI355X visible: the hot path has no CPU fallback')
        self.device = int(device)
        torch.cuda.set_device(self.device)
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        self._h = c_void_p()
        rc = lib().fv_create(self.device, c_void_p(stream), ctypes.byref(self._h))
        if rc != 0:
            raise FvError('fv_create failed (%d): %s' % (rc, lib().fv_last_error(None).decode()))

    def check(self, rc, what):
        if rc != 0:
            raise FvError('%s failed (%d): %s' % (what, rc, lib().fv_last_error(self._h).decode()))

    @property
    def handle(self):
        return self._h

    def set_stream(self, stream_ptr):
        self.check(lib().fv_set_stream(self._h, c_void_p(stream_ptr)), 'fv_set_stream')

    def close(self):
        if self._h:
            lib().fv_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ptr(t):
    """Raw device pointer of a contiguous torch tensor."""
    assert t.is_contiguous()
    return c_void_p(t.data_ptr())
