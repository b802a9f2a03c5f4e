"""Host-side (CPU) pieces of the product against the golden vectors and by property."""
import io
import os

import numpy as np
import pytest

from face_vijnana_yolov3_amd import data, weights


def test_encode_gt_matches_reference_golden(golden_dir):
    import pandas as pd
    g = np.load(os.path.join(golden_dir, 'gt_encoder.npz'))
    df = pd.read_csv(io.StringIO(str(g['csv'])))
    for k, f in enumerate(g['files']):
        rows = df[df['FILE'] == str(f)].iloc[:, 3:7].values
        gt = data.encode_gt(rows, int(g['hw'][k][0]), int(g['hw'][k][1]), 416, 13)
        np.testing.assert_array_equal(gt, g['gt'][k], err_msg=str(f))


def test_letterbox_geometry_and_pixels():
    for (h, w) in [(300, 500), (500, 300), (416, 416), (601, 1000), (123, 124), (1080, 1920)]:
        w_p, h_p, pt, pb, pl, pr = data.letterbox_geometry(h, w, 416)
        assert h_p + pt + pb == 416 and w_p + pl + pr == 416
        assert (pb - pt) in (0, 1) and (pr - pl) in (0, 1)      # extra row/col bottom/right
        img, geom = data.letterbox(np.full((h, w, 3), 128, np.uint8), 416)
        assert img.shape == (416, 416, 3) and geom[:2] == (h, w)
        inner = img[pt:416 - pb, pl:416 - pr]
        np.testing.assert_allclose(inner, 128 / 255, atol=1e-12)  # bicubic weights sum to 1
        assert img[:pt].sum() == 0 and img[:, :pl].sum() == 0
    # same size: identity
    rng = np.random.default_rng(0)
    raw = rng.integers(0, 256, (416, 416, 3), dtype=np.uint8)
    np.testing.assert_allclose(data.letterbox(raw, 416)[0], raw / 255, atol=1e-12)


def test_training_sequence_contract(tmp_path):
    df = data.make_synthetic_uccs(str(tmp_path), n_images=5, seed=3)
    hps = {'batch_size': 2, 'step': 1}
    seq = data.TrainingSequence(str(tmp_path), hps, {'image_size': 64, 'bb_info_c_size': 6})
    assert hps['step'] == 3 and len(seq) == 3                      # overwritten: ceil(5/2)
    assert seq.file_names == sorted(df['FILE'].unique())
    x, y = seq[0]
    assert x['input1'].shape == (2, 64, 64, 3) and y['output'].shape == (2, 2, 2, 6)
    x, y = seq[2]
    assert x['input1'].shape[0] == 1                                # short last batch
    assert y['output'][..., 0].sum() >= 1


def test_darknet_reader_matches_reference_golden_and_roundtrip(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, 'weight_reader.npz'))
    spec = [(0, 3, 3, 4), (1, 3, 4, 8), (3, 1, 8, 6)]
    layers, off, soff = [], 0, 0
    for idx, k, cin, cout in spec:
        d = dict(darknet_index=idx, ksize=k, cin=cin, cout=cout, has_bn=1, w_off=off)
        off += cout * k * k * cin
        d['gamma_off'] = off; off += cout
        d['beta_off'] = off; off += cout
        d['mean_off'] = soff; soff += cout
        d['var_off'] = soff; soff += cout
        layers.append(d)
    for tag in ('v2', 'v1'):
        p, s = weights.read_darknet_base(g[tag + '_file'].tobytes(), layers, off, soff)
        kw = weights.keras_weights(layers, p, s)
        for d in layers:
            i = d['darknet_index']
            np.testing.assert_array_equal(kw['conv_%d' % i][0], g['%s_conv_%d_0' % (tag, i)])
            for j in range(4):
                np.testing.assert_array_equal(kw['bnorm_%d' % i][j], g['%s_bnorm_%d_%d' % (tag, i, j)])
    path = str(tmp_path / 'synthetic.weights')
    weights.write_darknet_base(path, layers, p, s)
    p2, s2 = weights.read_darknet_base(path, layers, off, soff)
    np.testing.assert_array_equal(p, p2); np.testing.assert_array_equal(s, s2)
    with pytest.raises(ValueError):
        weights.read_darknet_base(open(path, 'rb').read()[:100], layers, off, soff)


def test_layer_table_without_gpu():
    from face_vijnana_yolov3_amd.engine import layer_table
    from oracle import net_oracle as no
    t = layer_table()
    ents, n, ns = no.param_layout()
    assert len(t) == 53 and t[-1]['role'] == 3 and t[-1]['beta_off'] + 6 == n
    assert [d['darknet_index'] for d in t[:-1]] == [e['idx'] for e in ents[:-1]]
    assert [d['in_div'] for d in t if d['stride'] == 2] == [1, 2, 4, 8, 16] and t[-1]['out_div'] == 32


def test_shard_files():
    from face_vijnana_yolov3_amd.parallel import shard_files
    names = ['f%d' % i for i in range(10)]
    parts = [shard_files(names, 4, r) for r in range(4)]
    assert sum(parts, []) == names and max(len(p) for p in parts) == 3
    assert shard_files(names[:2], 4, 3) == []


def test_cpu_letterbox_matches_oracle_statement():
    from oracle import host_oracle
    rng = np.random.default_rng(5)
    for (h, w) in [(37, 61), (80, 45), (64, 64), (20, 100)]:
        raw = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got, _ = data.letterbox(raw, 64)
        np.testing.assert_allclose(got, host_oracle.letterbox_pixels(raw, 64), rtol=0, atol=1e-12)


# ------------------------------------------------------------------------------------------ three-scale targets
def _random_faces(rng, h, w, n):
    fw = rng.uniform(6, w / 2.5, n); fh = rng.uniform(6, h / 2.5, n)
    fx = rng.uniform(1, w - fw - 1); fy = rng.uniform(1, h - fh - 1)
    rows = np.stack([fx, fy, fw, fh], 1)
    if n > 2:
        rows[1, 0] = 0.0          # a row the skip rule drops (fd.py:154-156)
    return rows


def test_three_scale_encoder_matches_the_oracle_restatement():
    from oracle import host_oracle
    rng = np.random.default_rng(11)
    for S in (416, 608, 96):
        for (h, w) in [(480, 640), (640, 480), (500, 500), (301, 1000), (1080, 607)]:
            rows = _random_faces(rng, h, w, 7)
            got = data.encode_gt_three_scale(rows, h, w, S)
            want = host_oracle.gt_encode_three_scale([tuple(r) for r in rows], h, w, S)
            for s in range(3):
                assert got[s].shape == (S // 32 << s, S // 32 << s, 18)
                np.testing.assert_allclose(got[s], want[s], rtol=0, atol=1e-12)
            assert sum(int((g.reshape(g.shape[0], g.shape[1], 3, 6)[..., 4] == 1).sum()) for g in got) <= 6


def test_three_scale_targets_decode_back_to_the_boxes_through_the_reference_decode():
    """encode -> (targets taken as network outputs, objectness / class logits +-12) -> the oracle restatement of the REFERENCE's
    decode_netout + correct_yolo_boxes (yolov3_detect.py:335-404, pinned by tests/golden/decode_netout*.npz) must give back
    every face: centre within the half-pixel clamp, size = the single-scale target's size, corners within 1 px of the truncated
    expectation -- and only anchors the reference's skip list keeps carry objects."""
    from oracle import host_oracle
    rng = np.random.default_rng(12)
    S = 416
    anchors = data.YOLO_ANCHORS
    for (h, w) in [(480, 640), (640, 480), (720, 1280)]:
        rows = _random_faces(rng, h, w, 5)
        tg = data.encode_gt_three_scale(rows, h, w, S)
        boxes = []
        for s in range(3):
            out = tg[s].reshape(tg[s].shape[0], tg[s].shape[1], 3, 6).copy()
            obj = out[..., 4] == 1
            for (sc, b) in [(0, 0), (0, 2), (1, 1), (2, 0), (2, 2)]:
                if sc == s:
                    assert not obj[..., b].any()                      # never assigned to an anchor the decode skips
            out[..., 4:] = np.where(out[..., 4:] > 0.5, 12.0, -12.0)
            boxes += host_oracle.decode_netout(out.reshape(tg[s].shape).astype(np.float32), anchors[s], s, 0.5, S, S)
        host_oracle.correct_yolo_boxes(boxes, S, S, S, S)
        m = max(h, w)
        _, _, pad_t, _, pad_l, _ = data.letterbox_geometry(h, w, S)
        ox, oy = (0, pad_t) if w >= h else (pad_l, 0)
        want = []
        for fx, fy, fw, fh in rows:
            if min(fx, fy, fw, fh) <= 0:
                continue
            x1, y1 = int(fx), int(fy); x2, y2 = x1 + int(fw) - 1, y1 + int(fh) - 1
            xc = (int(x1 / m * S) + ox + int(x2 / m * S) + ox) // 2; yc = (int(y1 / m * S) + oy + int(y2 / m * S) + oy) // 2
            bw, bh = (x2 - x1 + 1) / m * S, (y2 - y1 + 1) / m * S
            want.append((xc - bw / 2, yc - bh / 2, xc + bw / 2, yc + bh / 2))
        assert len(boxes) == len(want)
        for wb in want:
            d = [max(abs(b[k] - wb[k]) for k in range(4)) for b in boxes]
            assert min(d) <= 1.5, (wb, boxes)                         # int() truncation (< 1) + the half-pixel offset clamp


def test_training_sequence_three_scale_contract(tmp_path):
    data.make_synthetic_uccs(str(tmp_path), n_images=3, seed=4)
    seq = data.TrainingSequence(str(tmp_path), {'batch_size': 2, 'step': 1}, {'image_size': 64, 'bb_info_c_size': 6, 'head': 'three_scale'})
    x, y = seq[0]
    assert x['input1'].shape == (2, 64, 64, 3)
    assert [y['output%d' % s].shape for s in range(3)] == [(2, 2, 2, 18), (2, 4, 4, 18), (2, 8, 8, 18)]
    raws, gts = seq.get_raw(1)
    assert len(raws) == 1 and [g.shape for g in gts] == [(1, 2, 2, 18), (1, 4, 4, 18), (1, 8, 8, 18)]


def test_back_projection_on_arrays_equals_the_reference_expressions():
    """FaceDetector._project_back evaluates fd.py:700-710 for all boxes of an image at once; the values AND the text str() makes
    of them in the csv row must be the reference's per-box `np.min([...])` / `np.max([...])` expressions, bit for bit."""
    from face_vijnana_yolov3_amd.face_detection import FaceDetector
    from face_vijnana_yolov3_amd.postproc import BoundBox

    class F:
        image_size = 416
    S = 416
    rng = np.random.default_rng(0)
    for trial in range(200):
        h, w = [(768, 1024), (1024, 768), (600, 600), (333, 601), (1080, 1920)][trial % 5]
        pad_t, pad_l = int(rng.integers(0, 120)), int(rng.integers(0, 120))
        geom = (h, w, pad_t, 0, pad_l, 0)
        coords = rng.integers(-30, 500, (int(rng.integers(0, 70)), 4))
        boxes = [BoundBox(int(c[0]), int(c[1]), int(c[2]), int(c[3]), objness=0.5, classes=[0.7]) for c in coords]
        FaceDetector._project_back(F, boxes, geom)
        for b, c in zip(boxes, coords):
            xmin, ymin, xmax, ymax = [int(v) for v in c]
            if w >= h:
                ref = (np.min([xmin * w / S, w]), np.min([np.max([ymin - pad_t, 0]) * w / S, h]),
                       np.min([xmax * w / S, w]), np.min([np.max([ymax - pad_t, 0]) * w / S, h]))
            else:
                ref = (np.min([np.max([xmin - pad_l, 0]) * h / S, w]), np.min([ymin * h / S, h]),
                       np.min([np.max([xmax - pad_l, 0]) * h / S, w]), np.min([ymax * h / S, h]))
            got = (b.xmin, b.ymin, b.xmax, b.ymax)
            assert got == ref and [str(g) for g in got] == [str(r) for r in ref]
            assert str(b.xmax - b.xmin) == str(ref[2] - ref[0]) and str(b.ymax - b.ymin) == str(ref[3] - ref[1])


def test_default_loader_threads_respects_affinity_and_quota():
    from face_vijnana_yolov3_amd import face_detection as fdm
    n = fdm.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or n)
    assert 4 <= fdm.default_loader_threads() <= 32
    assert fdm.default_loader_threads() == max(4, min(32, n))


def test_default_eval_batch_fits_one_buffer_descriptor():
    """evaluate()/test() batch when hps.eval_batch_size is absent: 48 where the first layer's output (batch x S^2 x 32 floats) stays
    within the 2^29 elements a 2 GiB buffer descriptor addresses (fv_forward_infer refuses more), the largest multiple of 8 below
    that otherwise."""
    from face_vijnana_yolov3_amd.face_detection import default_eval_batch
    assert default_eval_batch(416) == 48 and default_eval_batch(320) == 48
    assert default_eval_batch(608) == 40
    for s in range(96, 4097, 32):
        b = default_eval_batch(s)
        assert b >= 1 and (b == 1 or b * s * s * 32 <= 1 << 29)
        assert b == 48 or (b + 8) * s * s * 32 > 1 << 29 or b < 8
