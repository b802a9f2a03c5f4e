"""ORACLE -- test infrastructure only.

Baseline / extended-sequential Huffman JPEG decoding restated in Python + numpy, bit for bit as libjpeg(-turbo) decodes with
its default settings (JDCT_ISLOW, fancy upsampling, integer YCbCr -> RGB) -- which is what the reference's
`skimage.io.imread` hands to FaceDetector (face_detection.py:112, 656, 798; scikit-image reads JPEG through Pillow/libjpeg).
Pinned by Pillow itself, which is importable here and on the GPU box: tests/test_jpeg_cpu.py requires this file to reproduce
`PIL.Image.open(f).convert('RGB')` exactly; the product's host entropy decoder and device kernels are then checked against both.

Algorithms restated from the IJG / libjpeg-turbo sources (third party, not in /root/reference): jdhuff.c (Huffman decoding),
jidctint.c `jpeg_idct_islow` (CONST_BITS 13, PASS1_BITS 2), jdsample.c `h2v1_fancy_upsample` / `h2v2_fancy_upsample`,
jdcolor.c `ycc_rgb_convert` (SCALEBITS 16), jdmainct.c edge handling (the last real row / column is replicated)."""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
                   62, 63])


class Unsupported(ValueError):
    pass


def parse(buf):
    """-> dict(width, height, comps=[dict(id, h, v, tq, td, ta)], qt={id: 64 ints natural order}, dc/ac huffman specs, ri, scan
    data bytes).  Raises Unsupported for progressive / arithmetic / 12-bit / CMYK files."""
    b = bytes(buf)
    if b[:2] != b'\xff\xd8':
        raise Unsupported('not a JPEG')
    p = 2
    info = dict(qt={}, dc={}, ac={}, ri=0, adobe=None)
    while p < len(b):
        if b[p] != 0xFF:
            raise Unsupported('marker expected')
        while b[p + 1] == 0xFF:
            p += 1
        m = b[p + 1]
        p += 2
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            continue
        n = (b[p] << 8) | b[p + 1]
        seg = b[p + 2:p + n]
        if m == 0xDB:
            q = 0
            while q < len(seg):
                pq, tq = seg[q] >> 4, seg[q] & 15
                if pq:
                    raise Unsupported('16-bit quantisation table')
                t = np.zeros(64, np.int32)
                t[ZIGZAG] = np.frombuffer(seg[q + 1:q + 65], np.uint8)
                info['qt'][tq] = t
                q += 65
        elif m in (0xC0, 0xC1):
            if seg[0] != 8:
                raise Unsupported('sample precision %d' % seg[0])
            info['height'], info['width'] = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4]
            info['comps'] = [dict(id=seg[6 + 3 * i], h=seg[7 + 3 * i] >> 4, v=seg[7 + 3 * i] & 15, tq=seg[8 + 3 * i]) for i in range(seg[5])]
        elif 0xC2 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise Unsupported('SOF%d (progressive / lossless / arithmetic) is not supported' % (m - 0xC0))
        elif m == 0xC4:
            q = 0
            while q < len(seg):
                tc, th = seg[q] >> 4, seg[q] & 15
                counts = list(seg[q + 1:q + 17])
                ns = sum(counts)
                info['ac' if tc else 'dc'][th] = (counts, list(seg[q + 17:q + 17 + ns]))
                q += 17 + ns
        elif m == 0xDD:
            info['ri'] = (seg[0] << 8) | seg[1]
        elif m == 0xEE and seg[:5] == b'Adobe':
            info['adobe'] = seg[11]
        elif m == 0xDA:
            ns = seg[0]
            for i in range(ns):
                cid, tt = seg[1 + 2 * i], seg[2 + 2 * i]
                for c in info['comps']:
                    if c['id'] == cid:
                        c['td'], c['ta'] = tt >> 4, tt & 15
            if ns != len(info['comps']):
                raise Unsupported('non-interleaved scans')
            info['scan'] = b[p + n:]
            break
        p += n
    if 'scan' not in info or 'comps' not in info:
        raise Unsupported('no frame / scan')
    nc = len(info['comps'])
    if nc not in (1, 3):
        raise Unsupported('%d components' % nc)
    if nc == 3 and info['adobe'] not in (None, 1):
        raise Unsupported('Adobe transform %r (not YCbCr)' % info['adobe'])
    return info


def _huff_table(counts, symbols):
    """code length, code -> symbol (canonical Huffman, jdhuff.c jpeg_make_d_derived_tbl)."""
    table = {}
    code, k = 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            table[(length, code)] = symbols[k]
            code += 1; k += 1
        code <<= 1
    return table


class _Bits(object):
    def __init__(self, data):
        self.d, self.p, self.acc, self.n = data, 0, 0, 0

    def bit(self):
        if self.n == 0:
            if self.p >= len(self.d):
                byte = 0                         # past the end: zeros (libjpeg inserts them with a warning)
            else:
                byte = self.d[self.p]; self.p += 1
                if byte == 0xFF:
                    nxt = self.d[self.p] if self.p < len(self.d) else 0xD9
                    if nxt == 0:
                        self.p += 1
                    else:                        # a marker: stay in front of it, feed zeros
                        self.p -= 1
                        byte = 0
            self.acc, self.n = byte, 8
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v

    def restart(self):
        self.n = 0
        while self.p + 1 < len(self.d) and not (self.d[self.p] == 0xFF and 0xD0 <= self.d[self.p + 1] <= 0xD7):
            self.p += 1
        self.p += 2


def entropy_decode(info):
    """-> list per component of int32 arrays [blocks_v][blocks_h][64] (natural order, NOT dequantised), the block grid padded
    to whole MCUs."""
    comps = info['comps']
    hmax, vmax = max(c['h'] for c in comps), max(c['v'] for c in comps)
    mcux, mcuy = -(-info['width'] // (8 * hmax)), -(-info['height'] // (8 * vmax))
    out = [np.zeros((mcuy * c['v'], mcux * c['h'], 64), np.int32) for c in comps]
    dct = {k: _huff_table(*v) for k, v in info['dc'].items()}
    act = {k: _huff_table(*v) for k, v in info['ac'].items()}
    br = _Bits(info['scan'])

    def sym(table):
        code = 0
        for length in range(1, 17):
            code = (code << 1) | br.bit()
            s = table.get((length, code))
            if s is not None:
                return s
        raise ValueError('bad Huffman code')

    def extend(v, n):
        return v if n == 0 or v >= (1 << (n - 1)) else v - (1 << n) + 1

    pred = [0] * len(comps)
    for m in range(mcux * mcuy):
        if info['ri'] and m and m % info['ri'] == 0:
            br.restart()
            pred = [0] * len(comps)
        my, mx = divmod(m, mcux)
        for ci, c in enumerate(comps):
            for by in range(c['v']):
                for bx in range(c['h']):
                    blk = out[ci][my * c['v'] + by, mx * c['h'] + bx]
                    n = sym(dct[c['td']])
                    pred[ci] += extend(br.bits(n), n)
                    blk[0] = pred[ci]
                    k = 1
                    while k < 64:
                        rs = sym(act[c['ta']])
                        r, s = rs >> 4, rs & 15
                        if s == 0:
                            if r != 15:
                                break
                            k += 16
                            continue
                        k += r
                        blk[ZIGZAG[k]] = extend(br.bits(s), s)
                        k += 1
    return out


# ----------------------------------------------------------------------------- reconstruction (vectorised)
def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def idct_islow(coef):
    """coef int [..., 64] dequantised, natural order -> uint8 [..., 8, 8] (jidctint.c jpeg_idct_islow)."""
    C = 13; P = 2
    F = dict(a=2446, b=3196, c=4433, d=6270, e=7373, f=9633, g=12299, h=15137, i=16069, j=16819, k=20995, l=25172)
    x = coef.astype(np.int64).reshape(coef.shape[:-1] + (8, 8))

    def one_d(v, shift, first):
        # v[..., k, :] = the k-th frequency of every column (pass 1) -- the caller transposes for pass 2
        z2, z3 = v[..., 2, :], v[..., 6, :]
        z1 = (z2 + z3) * F['c']
        t2 = z1 + z3 * (-F['h']); t3 = z1 + z2 * F['d']
        z2, z3 = v[..., 0, :], v[..., 4, :]
        t0 = (z2 + z3) << C; t1 = (z2 - z3) << C
        t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
        t0, t1, t2, t3 = v[..., 7, :], v[..., 5, :], v[..., 3, :], v[..., 1, :]
        z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
        z5 = (z3 + z4) * F['f']
        t0 = t0 * F['a']; t1 = t1 * F['j']; t2 = t2 * F['l']; t3 = t3 * F['g']
        z1 = z1 * (-F['e']); z2 = z2 * (-F['k']); z3 = z3 * (-F['i']) + z5; z4 = z4 * (-F['b']) + z5
        t0 = t0 + z1 + z3; t1 = t1 + z2 + z4; t2 = t2 + z2 + z3; t3 = t3 + z1 + z4
        rows = [t10 + t3, t11 + t2, t12 + t1, t13 + t0, t13 - t0, t12 - t1, t11 - t2, t10 - t3]
        return np.stack([_descale(r, shift) for r in rows], axis=-2)

    ws = one_d(x, C - P, True)                                  # pass 1: columns (index k runs down a column)
    res = one_d(np.swapaxes(ws, -1, -2), C + P + 3, False)      # pass 2: rows
    res = np.swapaxes(res, -1, -2)
    return np.clip(res + 128, 0, 255).astype(np.uint8)


def planes(info, blocks):
    """dequantise + IDCT -> one uint8 plane per component, cropped to the component's real size."""
    comps = info['comps']
    hmax, vmax = max(c['h'] for c in comps), max(c['v'] for c in comps)
    out = []
    for c, blk in zip(comps, blocks):
        px = idct_islow(blk * info['qt'][c['tq']])
        bv, bh = blk.shape[:2]
        plane = px.transpose(0, 2, 1, 3).reshape(bv * 8, bh * 8)
        w = -(-info['width'] * c['h'] // hmax); h = -(-info['height'] * c['v'] // vmax)
        out.append(plane[:h, :w])
    return out


def _h2v1_fancy(row):
    """jdsample.c h2v1_fancy_upsample on rows [..., w] -> [..., 2w]."""
    x = row.astype(np.int32)
    w = x.shape[-1]
    out = np.empty(x.shape[:-1] + (2 * w,), np.int32)
    if w == 1:
        out[..., 0] = x[..., 0]; out[..., 1] = x[..., 0]
        return out.astype(np.uint8)
    left = np.concatenate([x[..., :1], x[..., :-1]], -1)
    right = np.concatenate([x[..., 1:], x[..., -1:]], -1)
    out[..., 0::2] = (3 * x + left + 1) >> 2
    out[..., 1::2] = (3 * x + right + 2) >> 2
    out[..., 0] = x[..., 0]
    out[..., -1] = x[..., -1]
    return out.astype(np.uint8)


def _h2v2_fancy(plane):
    """jdsample.c h2v2_fancy_upsample: [h, w] -> [2h, 2w]; vertical neighbours replicate at the edges (jdmainct.c)."""
    x = plane.astype(np.int32)
    h, w = x.shape
    above = np.concatenate([x[:1], x[:-1]], 0)
    below = np.concatenate([x[1:], x[-1:]], 0)
    out = np.empty((2 * h, 2 * w), np.int32)
    for v, nb in ((0, above), (1, below)):
        col = 3 * x + nb                                         # thiscolsum
        last = np.concatenate([col[:, :1], col[:, :-1]], 1)
        nxt = np.concatenate([col[:, 1:], col[:, -1:]], 1)
        even = (3 * col + last + 8) >> 4
        odd = (3 * col + nxt + 7) >> 4
        if w == 1:
            even = odd = (4 * col + 8) >> 4
            odd = (4 * col + 7) >> 4
        else:
            even[:, 0] = (4 * col[:, 0] + 8) >> 4
            odd[:, -1] = (4 * col[:, -1] + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out.astype(np.uint8)


def upsample(info, pl):
    comps = info['comps']
    hmax, vmax = max(c['h'] for c in comps), max(c['v'] for c in comps)
    H, W = info['height'], info['width']
    out = []
    for c, p in zip(comps, pl):
        fh, fv = hmax // c['h'], vmax // c['v']
        if (fh, fv) == (1, 1):
            u = p
        elif (fh, fv) == (2, 1):
            u = _h2v1_fancy(p)
        elif (fh, fv) == (2, 2):
            u = _h2v2_fancy(p)
        else:
            raise Unsupported('sampling ratio %dx%d' % (fh, fv))
        out.append(u[:H, :W])
    return out


def ycc_to_rgb(y, cb, cr):
    """jdcolor.c ycc_rgb_convert (SCALEBITS 16, FIX(x) = int(x * 65536 + 0.5))."""
    fix = lambda v: int(v * 65536 + 0.5)
    half = 1 << 15
    yy = y.astype(np.int64); b = cb.astype(np.int64) - 128; r = cr.astype(np.int64) - 128
    R = yy + ((fix(1.40200) * r + half) >> 16)
    B = yy + ((fix(1.77200) * b + half) >> 16)
    G = yy + ((-fix(0.34414) * b + half - fix(0.71414) * r) >> 16)
    return np.clip(np.stack([R, G, B], -1), 0, 255).astype(np.uint8)


def reconstruct(info, blocks):
    pl = upsample(info, planes(info, blocks))
    if len(pl) == 1:
        return np.repeat(pl[0][..., None], 3, axis=2)
    return ycc_to_rgb(*pl)


def decode(buf):
    """JPEG bytes -> HxWx3 uint8, as PIL.Image.open(...).convert('RGB')."""
    info = parse(buf)
    return reconstruct(info, entropy_decode(info))
