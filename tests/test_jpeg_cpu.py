"""JPEG decode split host / device (reference `imread`, face_detection.py:112, 656, 798), CPU half:
  * oracle/jpeg_oracle.py (numpy restatement of libjpeg's default decode) == Pillow, bit for bit -- Pillow is the reference's
    actual reader (scikit-image's imread goes through it) and is importable here and on the GPU box, so it pins the oracle;
  * the product's host entropy decoder (fv_jpeg_parse / fv_jpeg_entropy_decode, C++ in libfv_hotpath.so, no GPU needed) produces
    exactly the oracle's coefficients, and oracle-reconstruct(product coefficients) == Pillow at sizes the Python Huffman loop
    would take minutes for;
  * what the decoder does not take (progressive, CMYK) is refused by parse() -> the loaders fall back to Pillow."""
import io

import numpy as np
import pytest
from PIL import Image

from oracle import jpeg_oracle as jo


def _jpeg(rng, h, w, sub, q, gray=False, ri=0, progressive=False, mode=None):
    lo = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    a = np.asarray(Image.fromarray(lo).resize((w, h), Image.BICUBIC)).astype(int) + rng.integers(-25, 25, (h, w, 3))
    im = Image.fromarray(np.clip(a, 0, 255).astype(np.uint8))
    if gray:
        im = im.convert('L')
    if mode:
        im = im.convert(mode)
    kw = {'restart_marker_blocks': ri} if ri else {}
    b = io.BytesIO()
    im.save(b, 'JPEG', quality=q, subsampling=sub, progressive=progressive, **kw)
    return b.getvalue()


CASES = [(45, 67, 0, 90, False, 0), (45, 67, 1, 75, False, 0), (45, 67, 2, 50, False, 0), (64, 64, 2, 95, False, 0), (33, 17, 2, 80, False, 0),
         (40, 56, 0, 85, True, 0), (50, 70, 2, 85, False, 3), (9, 9, 2, 90, False, 0), (1, 1, 2, 90, False, 0), (8, 31, 1, 100, False, 0),
         (100, 130, 1, 30, False, 2), (17, 200, 2, 5, False, 0)]


@pytest.mark.parametrize('h,w,sub,q,gray,ri', CASES)
def test_oracle_equals_pillow(h, w, sub, q, gray, ri):
    data = _jpeg(np.random.default_rng(h * 1000 + w), h, w, sub, q, gray, ri)
    want = np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))
    assert np.array_equal(jo.decode(data), want)


@pytest.mark.parametrize('h,w,sub,q,gray,ri', CASES + [(480, 640, 2, 90, False, 0), (601, 333, 1, 75, False, 7), (768, 1024, 2, 85, False, 0)])
def test_host_entropy_decoder_matches_oracle_and_pillow(h, w, sub, q, gray, ri):
    from face_vijnana_yolov3_amd import jpeg
    data = _jpeg(np.random.default_rng(h * 1000 + w + 1), h, w, sub, q, gray, ri)
    info = jpeg.parse(data)
    oi = jo.parse(data)
    assert info is not None and (info.width, info.height, info.ncomp, info.restart_interval) == (w, h, 1 if gray else 3, oi['ri'])
    coefs = jpeg.entropy_decode(data, info)
    blocks = jpeg.blocks_of(info, coefs)
    for c, comp in enumerate(oi['comps']):
        assert np.array_equal(np.asarray(info.qt[c][:], np.int32), oi['qt'][comp['tq']])
    if h * w <= 20000:                                      # the oracle's Python Huffman loop, coefficient by coefficient
        for mine, theirs in zip(blocks, jo.entropy_decode(oi)):
            assert np.array_equal(mine.astype(np.int32), theirs)
    rgb = jo.reconstruct(oi, [b.astype(np.int32) for b in blocks])
    assert np.array_equal(rgb, np.asarray(Image.open(io.BytesIO(data)).convert('RGB')))


def test_unsupported_files_are_refused():
    from face_vijnana_yolov3_amd import jpeg
    rng = np.random.default_rng(3)
    assert jpeg.parse(_jpeg(rng, 40, 40, 2, 90, progressive=True)) is None
    assert jpeg.parse(_jpeg(rng, 40, 40, 0, 90, mode='CMYK')) is None
    assert jpeg.parse(b'\x89PNG\r\n\x1a\n' + b'\0' * 64) is None
    good = _jpeg(rng, 40, 40, 2, 90)
    assert jpeg.parse(good[:100]) is None                   # truncated inside the headers
    info = jpeg.parse(good)
    with pytest.raises(AssertionError):
        jpeg.entropy_decode(good, info, np.empty(10, np.int16))
    with pytest.raises(jo.Unsupported):
        jo.parse(_jpeg(rng, 40, 40, 2, 90, progressive=True))


def test_damaged_files_never_crash_the_host_decoder():
    """Truncated files, flipped bytes anywhere, overwritten scan data: parse() refuses, entropy_decode() raises or returns
    (garbage) coefficients -- never a crash or a read past the buffer; a header claiming an absurd size is refused before any
    buffer is sized from it."""
    from face_vijnana_yolov3_amd import jpeg
    rng = np.random.default_rng(5)
    seen = {'decoded': 0, 'refused': 0, 'raised': 0}
    for trial in range(150):
        d = bytearray(_jpeg(rng, int(rng.integers(16, 120)), int(rng.integers(16, 120)), int(rng.integers(0, 3)), 85, ri=int(rng.integers(0, 2)) * 4))
        if trial % 3 == 0:
            d = d[:int(rng.integers(2, len(d)))]
        elif trial % 3 == 1:
            for _ in range(int(rng.integers(1, 8))):
                d[int(rng.integers(0, len(d)))] = int(rng.integers(0, 256))
        else:
            i = int(rng.integers(len(d) // 2, len(d)))
            d[i:i + 16] = bytes(rng.integers(0, 256, 16, dtype=np.uint8))
        info = jpeg.parse(bytes(d))
        if info is None:
            seen['refused'] += 1
            continue
        assert info.width * info.height <= jpeg.MAX_PIXELS
        try:
            jpeg.entropy_decode(bytes(d), info, np.zeros(int(info.total_coefs), np.int16))
            seen['decoded'] += 1
        except ValueError:
            seen['raised'] += 1
    assert seen['refused'] > 0 and seen['decoded'] + seen['raised'] > 0, seen
    # SOF0 with 65535 x 65535: 4.3 Gpixel
    good = bytearray(_jpeg(rng, 32, 32, 2, 85))
    k = good.find(b'\xff\xc0')
    good[k + 5:k + 9] = b'\xff\xff\xff\xff'
    assert jpeg.parse(bytes(good)) is None


def _flush_against_guard_page(data):
    """-> (pointer, keepalive): `data` copied so that its last byte is the last byte of a readable page and the next page is
    PROT_NONE -- a read one byte past the buffer faults instead of passing unnoticed (the CPU build's stand-in for ASan)."""
    import ctypes
    import mmap
    page = mmap.PAGESIZE
    npages = (len(data) + page - 1) // page + 1
    mm = mmap.mmap(-1, npages * page, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, prot=mmap.PROT_READ | mmap.PROT_WRITE)
    base = ctypes.addressof(ctypes.c_char.from_buffer(mm))
    libc = ctypes.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert libc.mprotect(base + (npages - 1) * page, page, 0) == 0          # PROT_NONE
    start = (npages - 1) * page - len(data)
    mm[start:start + len(data)] = data
    return base + start, mm


def test_parser_never_reads_past_the_buffer():
    """the fill-byte loop of parse() could read b[n] when a file ends in a run of 0xFF.  Every prefix of a real header
    and every tail of fill bytes is parsed flush against a PROT_NONE page."""
    import ctypes
    from face_vijnana_yolov3_amd import jpeg
    L = jpeg._fn()
    good = _jpeg(np.random.default_rng(11), 40, 56, 2, 90, ri=2)
    sos = good.find(b'\xff\xda')
    cases = [good[:k] for k in range(0, sos + 16)]
    cases += [b'\xff\xd8' + b'\xff' * k for k in range(0, 12)]
    cases += [good[:k] + b'\xff' * t for k in (2, 4, 20, 21, sos, sos + 2) for t in (1, 2, 3, 7)]
    for d in cases:
        if not d:
            continue
        p, keep = _flush_against_guard_page(d)
        info = jpeg.JpegInfo()
        rc = L.fv_jpeg_parse(ctypes.c_void_p(p), len(d), ctypes.byref(info))
        assert rc != 0 or len(d) > sos, (len(d), rc)
        del keep
    p, keep = _flush_against_guard_page(good)                      # the whole file, headers + scan, both entry points
    info = jpeg.JpegInfo()
    assert L.fv_jpeg_parse(ctypes.c_void_p(p), len(good), ctypes.byref(info)) == 0
    out = np.zeros(int(info.total_coefs), np.int16)
    assert L.fv_jpeg_entropy_decode(ctypes.c_void_p(p), len(good), ctypes.c_void_p(out.ctypes.data), out.size) == 0
    assert np.array_equal(out, jpeg.entropy_decode(good, jpeg.parse(good)))


def test_loader_falls_back_to_pillow_on_damaged_scan_data(tmp_path):
    """a scan-level failure of the host Huffman decoder (bad code, run past the block) used to raise out of the
    loader thread and abort train() / test(); libjpeg -- the reference's reader behind imread (fd.py:112, 798) -- only warns and
    still returns an image.  The batch must go through Pillow instead."""
    import pandas as pd
    from face_vijnana_yolov3_amd import data, jpeg
    from face_vijnana_yolov3_amd.face_detection import BatchFeeder
    rng = np.random.default_rng(21)
    good = _jpeg(rng, 96, 128, 2, 90)
    sos = good.find(b'\xff\xda')
    bad = None
    for trial in range(400):                                        # a corruption the product refuses and Pillow still decodes
        d = bytearray(good)
        i = int(rng.integers(sos + 14, len(d) - 2))
        d[i] = int(rng.integers(0, 255))
        d = bytes(d)
        info = jpeg.parse(d)
        if info is None:
            continue
        try:
            jpeg.entropy_decode(d, info)
            continue
        except ValueError:
            pass
        try:
            Image.open(io.BytesIO(d)).convert('RGB')
        except OSError:
            continue
        bad = d
        break
    assert bad is not None, 'no single-byte corruption found that the host decoder refuses and Pillow decodes'
    root = str(tmp_path)
    open(tmp_path / 'a.jpg', 'wb').write(good)
    open(tmp_path / 'b.jpg', 'wb').write(bad)
    rows = [[0, 'a.jpg', 1, 10.0, 12.0, 30.0, 30.0], [1, 'b.jpg', 1, 20.0, 22.0, 25.0, 35.0]]
    pd.DataFrame(rows, columns=data.CSV_COLUMNS).to_csv(tmp_path / 'training.csv', index=False)
    hps = dict(batch_size=2, step=1, lr=1e-4, beta_1=0.9, beta_2=0.99, decay=0.0)
    seq = data.TrainingSequence(root, hps, {'image_size': 96, 'bb_info_c_size': 6})
    f = BatchFeeder(seq, 1, 0, threads=2)
    try:
        packed, yt, weight, _ = f.load(0)
    finally:
        f.close()
    assert not isinstance(packed[0], str), 'the damaged batch must not arrive as JPEG coefficients'
    buf, offs, hw = packed
    assert list(hw) == [96, 128, 96, 128] and tuple(yt.shape) == (2, 3, 3, 6)
    want = np.asarray(Image.open(io.BytesIO(bad)).convert('RGB')).reshape(-1)
    assert np.array_equal(buf.numpy()[offs[1]:offs[1] + want.size], want)
