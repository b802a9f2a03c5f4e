"""Pin the oracle to golden vectors minted from the reference's own functions
(tests/golden/make_golden.py).  CPU only."""
import io
import os
import warnings

import numpy as np
import pytest

from oracle import host_oracle, postproc


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_detect_postproc_matches_reference(golden_dir):
    g = _load(golden_dir, 'detect_cases.npz')
    head, meta, out, cnt, names = g['head'], g['meta'], g['out'], g['out_count'], g['names']
    assert len(head) >= 100
    for k in range(len(head)):
        cth, ith, nc, _ = meta[k]
        r = postproc.detect_postproc(head[k:k + 1], 416, cth, ith, int(nc))
        c = int(r['count'][0])
        assert c == cnt[k], names[k]
        # integer boxes and the NMS/top-k index set + order: bit-exact
        np.testing.assert_array_equal(r['boxes'][0, :c], out[k, :c, :4].astype(np.int32), err_msg=str(names[k]))
        np.testing.assert_array_equal(r['cell'][0, :c], out[k, :c, 4].astype(np.int32), err_msg=str(names[k]))
        # float32 objness/score: reference uses NumPy's SIMD float32 exp; tolerance 2 ulp (2.4e-7 rel)
        np.testing.assert_allclose(r['obj'][0, :c], out[k, :c, 5], rtol=2.5e-7, atol=0)
        np.testing.assert_allclose(r['score'][0, :c], out[k, :c, 6], rtol=4e-7, atol=0)
        # ascending order (the reference keeps the LOWEST num_cands)
        assert np.all(np.diff(r['score'][0, :c]) >= 0)


def test_detect_edge_case_counts(golden_dir):
    g = _load(golden_dir, 'detect_cases.npz')
    names = list(g['names']); cnt = g['out_count']
    assert cnt[names.index('none')] == 0
    assert cnt[names.index('at_threshold')] == 2
    assert cnt[names.index('more_than_60')] == 60
    assert cnt[names.index('single')] == 1
    assert cnt[names.index('th_025')] == 10 and cnt[names.index('th_075')] == 5


def test_bbox_iou_bit_exact(golden_dir):
    g = _load(golden_dir, 'iou_cases.npz')
    got = postproc.bbox_iou(g['a'], g['b'])
    assert np.isnan(g['iou']).sum() >= 50  # zero-area pairs: 0/0 -> nan, never suppresses
    assert np.array_equal(got, g['iou'], equal_nan=True)
    assert postproc.bbox_iou(np.array([[0, 0, 10, 10]]), np.array([[1, 1, 11, 11]]))[0] == 0.680672268907563


def test_gt_encoder_matches_reference(golden_dir):
    import pandas as pd
    g = _load(golden_dir, 'gt_encoder.npz')
    df = pd.read_csv(io.StringIO(str(g['csv'])))
    files = [str(f) for f in g['files']]
    assert files == sorted(df['FILE'].unique())
    hw = g['hw']
    batches = host_oracle.training_batches(files, int(g['batch_size']))
    assert len(batches) == int(g['step'])
    assert [len(b) for b in batches] == list(g['batch_counts'])
    for k, f in enumerate(files):
        rows = df[df['FILE'] == f][['FACE_X', 'FACE_Y', 'FACE_WIDTH', 'FACE_HEIGHT']].values.tolist()
        gt = host_oracle.gt_encode_image(rows, int(hw[k][0]), int(hw[k][1]))
        np.testing.assert_array_equal(gt, g['gt'][k], err_msg=f)
    # the letterboxed image is always S x S
    for (h, w) in hw:
        w_p, h_p, pt, pb, pl, pr = host_oracle.letterbox_geometry(int(h), int(w), 416)
        assert h_p + pt + pb == 416 and w_p + pl + pr == 416
    assert np.all(g['image_shapes'][:, 1:] == [416, 416, 3])


def test_weight_reader_matches_reference(golden_dir):
    g = _load(golden_dir, 'weight_reader.npz')
    layers = [(0, (3, 3, 3, 4), True), (1, (3, 3, 4, 8), True), (3, (1, 1, 8, 6), True), (81, (1, 1, 6, 5), False)]
    for tag in ('v2', 'v1'):
        buf = g[tag + '_file'].tobytes()
        assert host_oracle.darknet_header_len(buf) == (20 if tag == 'v2' else 16)
        out, used = host_oracle.read_darknet_weights(buf, layers)
        assert used == int(g[tag + '_offset'])
        for name, arrs in out.items():
            for k, a in enumerate(arrs):
                np.testing.assert_array_equal(a, g['%s_%s_%d' % (tag, name, k)], err_msg=name)


def coco80_netouts(g):
    """Rebuild the dense head outputs of decode_netout_coco80.npz from its sparse form."""
    outs = []
    for s, gsz in enumerate((13, 26, 52)):
        no = np.zeros((gsz, gsz, g['vals_%d' % s].shape[1]), np.float32)
        no.reshape(gsz, gsz, 3, -1)[..., 4] = g['background_obj']
        c = g['cells_%d' % s]
        no[c[:, 0], c[:, 1]] = g['vals_%d' % s]
        outs.append(no)
    return outs


@pytest.mark.parametrize('fixture', ['decode_netout.npz', 'decode_netout_coco80.npz'])
def test_decode_netout_and_nms_secondary(golden_dir, fixture):
    g = _load(golden_dir, fixture)
    netouts = coco80_netouts(g) if 'coco80' in fixture else [g['netout_%d' % s] for s in range(3)]
    rows = []
    for s in range(3):
        rows += host_oracle.decode_netout(netouts[s], list(g['anchors'][s]), s, 0.5, 416, 416)
    pre = np.array(rows, np.float64)
    assert pre.shape == g['pre'].shape
    np.testing.assert_allclose(pre, g['pre'], rtol=1e-6, atol=1e-7)
    ih, iw = [int(v) for v in g['image_hw']]
    host_oracle.correct_yolo_boxes(rows, ih, iw, 416, 416)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        host_oracle.do_nms(rows, 0.5)
    post = np.array(rows, np.float64)
    np.testing.assert_array_equal(post[:, :4], g['post'][:, :4])
    np.testing.assert_array_equal(post[:, 5:] == 0, g['post'][:, 5:] == 0)
