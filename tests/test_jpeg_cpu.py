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
