"""JPEG input without a host-side image: Huffman decoding on host threads, everything after it on the device
(csrc/jpeg.hip; replaces `imread` of the reference's loaders, face_detection.py:112, 656, 798, for baseline JPEGs).

    info = parse(data)                       # None: not a file this decoder takes -> the caller uses Pillow for it
    entropy_decode(data, info, out_int16)    # quantised coefficients into a (pinned) host buffer; releases the GIL
    rgb = reconstruct_batch(ctx, ...)        # dequantise + IDCT + chroma upsampling + YCbCr->RGB for a whole batch

The pixels are bit-identical to Pillow's (tests/test_jpeg_cpu.py, tests/test_jpeg_gpu.py)."""
import ctypes

import numpy as np

from ._lib import lib, ptr


class JpegInfo(ctypes.Structure):
    """fv_jpeg_info of include/fv_hotpath.h."""
    _fields_ = [('width', ctypes.c_int32), ('height', ctypes.c_int32), ('ncomp', ctypes.c_int32), ('hmax', ctypes.c_int32),
                ('vmax', ctypes.c_int32), ('restart_interval', ctypes.c_int32),
                ('h', ctypes.c_int32 * 3), ('v', ctypes.c_int32 * 3), ('blocks_w', ctypes.c_int32 * 3), ('blocks_h', ctypes.c_int32 * 3),
                ('coef_off', ctypes.c_int64 * 3), ('total_coefs', ctypes.c_int64), ('qt', (ctypes.c_uint16 * 64) * 3)]


class JpegDesc(ctypes.Structure):
    """fv_jpeg_desc of include/fv_hotpath.h."""
    _fields_ = [('width', ctypes.c_int32), ('height', ctypes.c_int32), ('ncomp', ctypes.c_int32), ('hmax', ctypes.c_int32),
                ('vmax', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('blocks_w', ctypes.c_int32 * 3), ('blocks_h', ctypes.c_int32 * 3),
                ('coef_off', ctypes.c_int64 * 3), ('plane_off', ctypes.c_int64 * 3), ('rgb_off', ctypes.c_int64),
                ('qt', (ctypes.c_uint16 * 64) * 3)]


assert ctypes.sizeof(JpegInfo) == 488 and ctypes.sizeof(JpegDesc) == 488


def _fn():
    L = lib()
    if not getattr(L, '_jpeg_declared', False):
        L.fv_jpeg_parse.restype = ctypes.c_int
        L.fv_jpeg_parse.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(JpegInfo)]
        L.fv_jpeg_entropy_decode.restype = ctypes.c_int
        L.fv_jpeg_entropy_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int64]
        L.fv_jpeg_plane_bytes.restype = ctypes.c_int64
        L.fv_jpeg_plane_bytes.argtypes = [ctypes.POINTER(JpegInfo)]
        L.fv_jpeg_reconstruct_batch.restype = ctypes.c_int
        L.fv_jpeg_reconstruct_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
        L._jpeg_declared = True
    return L


# Pillow refuses images of more than 2 x Image.MAX_IMAGE_PIXELS pixels; a header claiming more is damaged or hostile, and the
# coefficient buffer for it would be page-locked before anything could notice
MAX_PIXELS = 178956970


def parse(data):
    """JPEG bytes -> JpegInfo, or None when the file is not one this decoder takes (progressive, arithmetic-coded, CMYK, 12-bit,
    unusual sampling, damaged header)."""
    info = JpegInfo()
    data = data if isinstance(data, bytes) else bytes(data)
    rc = _fn().fv_jpeg_parse(ctypes.c_char_p(data), len(data), ctypes.byref(info))
    if rc != 0 or info.width * info.height > MAX_PIXELS:
        return None          # implausible size (damaged header): left to Pillow, whose decompression-bomb check raises
    return info


def entropy_decode(data, info, out=None):
    """Huffman-decode the scan into `out` (int16 numpy array / memory of at least info.total_coefs elements; allocated when
    None).  ctypes releases the GIL for the duration of the call: a thread pool decodes in parallel.  -> the int16 array."""
    n = int(info.total_coefs)
    if out is None:
        out = np.empty(n, np.int16)
    assert out.dtype == np.int16 and out.size >= n and out.flags['C_CONTIGUOUS']
    data = data if isinstance(data, bytes) else bytes(data)
    rc = _fn().fv_jpeg_entropy_decode(ctypes.c_char_p(data), len(data), ctypes.c_void_p(out.ctypes.data), out.size)
    if rc != 0:
        raise ValueError('fv_jpeg_entropy_decode failed (%d): damaged scan data' % rc)
    return out


def blocks_of(info, coefs):
    """(test aid) the flat coefficient array -> one [blocks_h][blocks_w][64] view per component."""
    return [np.asarray(coefs[int(info.coef_off[c]):int(info.coef_off[c]) + info.blocks_w[c] * info.blocks_h[c] * 64]).reshape(
        info.blocks_h[c], info.blocks_w[c], 64) for c in range(info.ncomp)]


class BatchPlan(object):
    """Host-side layout of one batch: where each image's coefficients, component planes and RGB pixels live."""

    def __init__(self, infos):
        self.infos = infos
        self.n = len(infos)
        self.descs = (JpegDesc * self.n)()
        coef = plane = rgb = 0
        self.coef_off, self.rgb_off, self.hw = [], [], []
        self.max_blocks = self.max_pixels = 1
        for i, I in enumerate(infos):
            d = self.descs[i]
            d.width, d.height, d.ncomp, d.hmax, d.vmax = I.width, I.height, I.ncomp, I.hmax, I.vmax
            nb = 0
            for c in range(I.ncomp):
                d.blocks_w[c], d.blocks_h[c] = I.blocks_w[c], I.blocks_h[c]
                d.coef_off[c] = coef + I.coef_off[c]
                d.plane_off[c] = plane
                plane += I.blocks_w[c] * I.blocks_h[c] * 64
                nb += I.blocks_w[c] * I.blocks_h[c]
                for k in range(64):
                    d.qt[c][k] = I.qt[c][k]
            plane = (plane + 15) & ~15
            d.rgb_off = rgb
            self.coef_off.append(coef); self.rgb_off.append(rgb); self.hw += [I.height, I.width]
            coef += int(I.total_coefs)
            rgb += I.height * I.width * 3
            self.max_blocks = max(self.max_blocks, nb); self.max_pixels = max(self.max_pixels, I.height * I.width)
        self.total_coefs, self.plane_bytes, self.rgb_bytes = coef, plane, rgb


def reconstruct_batch(ctx, plan, coefs_dev, device):
    """coefficients of the batch (int16 CUDA tensor laid out by `plan`) -> packed RGB uint8 CUDA tensor (image i at
    plan.rgb_off[i], hw plan.hw[2i:2i+2]): what fv_letterbox_batch takes as `packed`."""
    import torch
    descs = torch.frombuffer(bytearray(bytes(plan.descs)), dtype=torch.uint8).to(device, non_blocking=True)
    planes = torch.empty(max(plan.plane_bytes, 16), dtype=torch.uint8, device=device)
    rgb = torch.empty(max(plan.rgb_bytes, 16), dtype=torch.uint8, device=device)
    rc = _fn().fv_jpeg_reconstruct_batch(ctx.handle, ptr(coefs_dev), ptr(descs), plan.n, ptr(planes), ptr(rgb), plan.max_blocks, plan.max_pixels)
    ctx.check(rc, 'fv_jpeg_reconstruct_batch')
    return rgb
