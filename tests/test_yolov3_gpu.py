"""Secondary path on the GPU: the full three-scale YOLOv3 forward against the
torch-CPU oracle, and decode_netout + correct_yolo_boxes + do_nms against the golden vectors minted
from the reference's own functions (tests/golden/decode_netout.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def model():
    from face_vijnana_yolov3_amd.yolov3 import Yolov3
    return Yolov3(0, out_channels=255)


def test_layout_matches_oracle(model):
    from oracle import net_oracle as no
    ents, n, ns = no.yolov3_layout(255)
    assert model.n_params == n == 61949149 and model.n_state == ns == 52608 and len(model.layers) == len(ents) == 75
    for d, e in zip(model.layers, ents):
        assert (d['darknet_index'], d['ksize'], d['cin'], d['cout'], d['w_off']) == (e['idx'], e['k'], e['cin'], e['cout'], e['w_off'])
        if e['has_bn']:
            assert (d['gamma_off'], d['beta_off'], d['mean_off'], d['var_off']) == (e['gamma_off'], e['beta_off'], e['mean_off'], e['var_off'])
        else:
            assert d['beta_off'] == e['bias_off'] and d['role'] == 5


@pytest.mark.parametrize('B,S', [(1, 96), (2, 64)])
def test_three_scale_forward_matches_oracle(model, B, S):
    from oracle import net_oracle as no
    p64, s64 = no.yolov3_init(11, 255, torch.float64)
    x = torch.rand((B, S, S, 3), dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    r64 = no.yolov3_forward(p64, s64, x)
    r32 = no.yolov3_forward(p64.float(), s64.float(), x.float())
    model.set_params(p64.float(), s64.float())
    ys = model.predict_device(x.float())
    torch.cuda.synchronize()
    for y, a, b, name in zip(ys, r64, r32, ('yolo_82', 'yolo_94', 'yolo_106')):
        assert tuple(y.shape) == tuple(a.shape)
        e_gpu = (y.cpu().double() - a).abs().max().item()
        e_cpu = (b.double() - a).abs().max().item()
        assert e_gpu <= 4 * e_cpu + 1e-6 * max(1.0, a.abs().max().item()), (name, e_gpu, e_cpu)


def test_darknet_full_weights_roundtrip(model, tmp_path):
    """Full-model .weights ordering (conv 0..105, yd.py:90-121): write a synthetic file in that
    order from the oracle layout and read it back through Yolov3.load_darknet."""
    import struct
    from oracle import net_oracle as no
    ents, n, ns = no.yolov3_layout(255)
    p, s = no.yolov3_init(5, 255, torch.float32)
    p, s = p.numpy(), s.numpy()
    path = str(tmp_path / 'y.weights')
    with open(path, 'wb') as f:
        f.write(struct.pack('iii', 0, 2, 0) + struct.pack('q', 0))
        for e in sorted(ents, key=lambda e: e['idx']):
            k, cin, cout = e['k'], e['cin'], e['cout']
            if e['has_bn']:
                for arr, o in ((p, e['beta_off']), (p, e['gamma_off']), (s, e['mean_off']), (s, e['var_off'])):
                    f.write(arr[o:o + cout].tobytes())
            else:
                f.write(p[e['bias_off']:e['bias_off'] + cout].tobytes())
            f.write(p[e['w_off']:e['w_off'] + cout * k * k * cin].reshape(cout, k, k, cin).transpose(0, 3, 1, 2).tobytes())
    used = model.load_darknet(path)
    assert used == n + ns
    assert np.array_equal(model.params.cpu().numpy(), p) and np.array_equal(model.state.cpu().numpy(), s)


def test_decode_nms_matches_reference_golden(model, golden_dir):
    from face_vijnana_yolov3_amd.yolov3 import decode_nms
    g = np.load(os.path.join(golden_dir, 'decode_netout.npz'))
    ys = [torch.from_numpy(g['netout_%d' % s]).cuda() for s in range(3)]
    ih, iw = [int(v) for v in g['image_hw']]
    res = decode_nms(model.ctx, ys[0], ys[1], ys[2], (ih, iw), (416, 416), g['anchors'].tolist(), 0.5, 0.5)
    post = g['post']
    n = post.shape[0]
    assert res['boxes'].shape[0] == n                                   # same candidates, same order
    np.testing.assert_allclose(res['objness'].cpu().numpy(), post[:, 4], rtol=3e-7, atol=0)
    # integer corners: float32 chain with a 1-ulp different exp may flip an int() truncation
    d = np.abs(res['boxes'].cpu().numpy().astype(np.int64) - post[:, :4].astype(np.int64))
    assert d.max() <= 1 and (d == 0).mean() >= 0.995, (d.max(), (d == 0).mean())
    cls = res['classes'].cpu().numpy()
    same_zero = (cls == 0) == (post[:, 5:] == 0)                        # NMS suppression pattern per class
    assert same_zero.mean() >= 0.995, same_zero.mean()
    nz = (cls != 0) & (post[:, 5:] != 0)
    np.testing.assert_allclose(cls[nz], post[:, 5:][nz], rtol=3e-7, atol=0)
    # do_nms really suppressed something in this fixture
    assert ((g['pre'][:, 5:] != 0) & (post[:, 5:] == 0)).sum() > 10


def test_decode_nms_properties_at_full_size(model):
    """416 input, COCO shape (80 classes): size-independent properties of the per-class NMS."""
    from face_vijnana_yolov3_amd.yolov3 import decode_nms
    from oracle import postproc as opp
    rng = np.random.default_rng(1)
    ys = []
    for gsz in (13, 26, 52):
        a = rng.normal(0, 1.0, (gsz, gsz, 255)).astype(np.float32)
        a.reshape(gsz, gsz, 3, 85)[..., 4] -= 1.5
        a.reshape(gsz, gsz, 3, 85)[..., 2:4] *= 0.3
        ys.append(torch.from_numpy(a).cuda())
    r1 = decode_nms(model.ctx, *ys, (1440, 1920), (416, 416), obj_thresh=0.5, nms_thresh=0.5)
    r2 = decode_nms(model.ctx, *ys, (1440, 1920), (416, 416), obj_thresh=0.5, nms_thresh=0.5)
    for k in r1:
        assert torch.equal(r1[k], r2[k])                                 # idempotent / deterministic
    n = r1['boxes'].shape[0]
    assert 100 < n <= 13 * 13 + 2 * 26 * 26 + 52 * 52
    b = r1['boxes'].cpu().numpy(); c = r1['classes'].cpu().numpy()
    for cl in (0, 17, 79):                                               # survivors of a class never overlap >= th
        idx = np.nonzero(c[:, cl] > 0)[0][:400]
        ii, jj = np.triu_indices(len(idx), 1)
        iou = opp.bbox_iou(b[idx][ii], b[idx][jj])
        assert not np.any(iou >= 0.5)
