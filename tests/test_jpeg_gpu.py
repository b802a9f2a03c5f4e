"""JPEG decode split host / device, GPU half (csrc/jpeg.hip: jpeg_idct_kernel, jpeg_color_kernel through the C ABI): the RGB
pixels the device reconstructs from the host's Huffman-decoded coefficients are bit-identical to Pillow's -- the reference's
reader (`imread`, face_detection.py:112, 656, 798) -- for every sampling mode, odd sizes, restart intervals and mixed batches;
the letterboxed network input is therefore the same tensor as through the Pillow path, and FaceDetector.test() writes the
same rows either way."""
import io
import os

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu

from test_jpeg_cpu import CASES, _jpeg  # noqa: E402  (tests/ is on sys.path under pytest's rootdir conftest)


@pytest.fixture(scope='module')
def ctx():
    from face_vijnana_yolov3_amd._lib import Context
    return Context(0)


def _device_decode(ctx, datas):
    from face_vijnana_yolov3_amd import jpeg
    infos = [jpeg.parse(d) for d in datas]
    assert all(i is not None for i in infos)
    plan = jpeg.BatchPlan(infos)
    coefs = np.empty(plan.total_coefs, np.int16)
    for i, d in enumerate(datas):
        jpeg.entropy_decode(d, infos[i], coefs[plan.coef_off[i]:plan.coef_off[i] + int(infos[i].total_coefs)])
    rgb = jpeg.reconstruct_batch(ctx, plan, torch.from_numpy(coefs).cuda(), torch.device('cuda', 0)).cpu().numpy()
    return [rgb[plan.rgb_off[i]:plan.rgb_off[i] + I.height * I.width * 3].reshape(I.height, I.width, 3) for i, I in enumerate(infos)], plan


def test_device_pixels_equal_pillow_in_one_mixed_batch(ctx):
    rng = np.random.default_rng(7)
    cases = CASES + [(480, 640, 2, 90, False, 0), (601, 333, 1, 75, False, 7), (768, 1024, 2, 85, False, 0), (1080, 1920, 2, 92, False, 0),
                     (333, 601, 0, 60, True, 0)]
    datas = [_jpeg(rng, *c) for c in cases]
    got, _plan = _device_decode(ctx, datas)
    for c, d, g in zip(cases, datas, got):
        want = np.asarray(Image.open(io.BytesIO(d)).convert('RGB'))
        assert g.shape == want.shape and np.array_equal(g, want), (c, int(np.abs(g.astype(int) - want.astype(int)).max()))


def test_letterbox_from_coefficients_equals_letterbox_from_pillow(ctx):
    from face_vijnana_yolov3_amd import jpeg
    from face_vijnana_yolov3_amd.postproc import letterbox_batch_device
    rng = np.random.default_rng(8)
    datas = [_jpeg(rng, h, w, sub, 88) for (h, w, sub) in [(480, 640, 2), (640, 480, 1), (300, 300, 0), (721, 1283, 2)]]
    infos = [jpeg.parse(d) for d in datas]
    plan = jpeg.BatchPlan(infos)
    coefs = torch.empty(plan.total_coefs, dtype=torch.int16).pin_memory()
    for i, d in enumerate(datas):
        jpeg.entropy_decode(d, infos[i], coefs.numpy()[plan.coef_off[i]:plan.coef_off[i] + int(infos[i].total_coefs)])
    dev = torch.device('cuda', 0)
    x1, g1 = letterbox_batch_device(ctx, None, 416, dev, packed=('jpeg', coefs, plan))
    raws = [np.asarray(Image.open(io.BytesIO(d)).convert('RGB')) for d in datas]
    x2, g2 = letterbox_batch_device(ctx, raws, 416, dev)
    assert g1 == g2 and torch.equal(x1, x2)


def test_facedetector_test_rows_do_not_depend_on_the_decoder(tmp_path, monkeypatch):
    from face_vijnana_yolov3_amd import face_detection
    from face_vijnana_yolov3_amd.face_detection import FaceDetector
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(face_detection, 'DEBUG', False)
    root = str(tmp_path / 'imgs'); os.makedirs(root)
    rng = np.random.default_rng(9)
    for k, (h, w, sub) in enumerate([(480, 640, 2), (640, 480, 1), (416, 416, 0), (300, 520, 2), (520, 300, 2)]):
        open(os.path.join(root, 'img_%d.jpg' % k), 'wb').write(_jpeg(rng, h, w, sub, 90))
    Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)).save(os.path.join(root, 'img_9.jpg'), progressive=True)   # -> Pillow
    conf = {'mode': 'test', 'raw_data_path': root, 'test_path': root, 'output_file_path': os.path.join(root, 'a.csv'), 'multi_gpu': False,
            'num_gpus': 1, 'yolov3_base_model_load': False, 'model_loading': False,
            'hps': {'lr': 1e-4, 'beta_1': 0.99, 'beta_2': 0.99, 'decay': 0.0, 'epochs': 1, 'step': 1, 'batch_size': 2, 'face_conf_th': 0.5,
                    'nms_iou_th': 0.5, 'num_cands': 60, 'eval_batch_size': 3},
            'nn_arch': {'image_size': 416, 'bb_info_c_size': 6}}
    fd = FaceDetector(conf)
    d = fd.model.layers[-1]
    fd.model.params[d['w_off']:d['beta_off']] *= 0.02                 # a head that depends on the pixels and fires on some cells
    fd.model.params[d['beta_off']] = 0.4; fd.model.params[d['beta_off'] + 5] = 0.4
    fd.test()
    rows_dev = open(conf['output_file_path']).read()
    conf['hps']['device_jpeg'] = False
    fd.conf = dict(conf, output_file_path=os.path.join(root, 'b.csv'))
    fd.test()
    assert rows_dev == open(os.path.join(root, 'b.csv')).read() and len(rows_dev.splitlines()) > 0


def test_training_feeder_through_coefficients_equals_pillow(tmp_path):
    """BatchFeeder with hps.device_jpeg (default) and without: the same letterboxed batch and the same targets."""
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.engine import Engine
    from face_vijnana_yolov3_amd.face_detection import BatchFeeder
    from face_vijnana_yolov3_amd.postproc import letterbox_batch_device
    root = str(tmp_path)
    data.make_synthetic_uccs(root, n_images=4, seed=5)
    eng = Engine(0)
    xs = []
    for dj in (True, False):
        seq = data.TrainingSequence(root, {'batch_size': 4, 'step': 1, 'device_jpeg': dj}, {'image_size': 416, 'bb_info_c_size': 6})
        f = BatchFeeder(seq, 1, 0, 4)
        packed, y, weight, (_f, slot) = f.load(0)
        assert (isinstance(packed[0], str) and packed[0] == 'jpeg') == dj
        x, _ = letterbox_batch_device(eng.ctx, None, 416, eng.dev, packed=packed)
        xs.append((x.clone(), y.clone()))
        f.close()
    assert torch.equal(xs[0][0], xs[1][0]) and torch.equal(xs[0][1], xs[1][1])


def test_pipelined_input_path_equals_the_serial_one(tmp_path):
    """run_pipelined (batch k+1 staged on its own stream while step k computes, losses reported one step late) against
    train_on_item (everything on the compute stream): same batches in the same order -> the same losses and, up to the float-atomic
    order inside dW, the same parameters after three Adam steps."""
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.engine import Engine
    from face_vijnana_yolov3_amd.face_detection import BatchFeeder, run_pipelined, train_on_item
    from face_vijnana_yolov3_amd.parallel import DataParallelTrainer
    root = str(tmp_path)
    data.make_synthetic_uccs(root, n_images=6, seed=6)
    hp = {'batch_size': 2, 'step': 1, 'lr': 1e-4, 'beta_1': 0.99, 'beta_2': 0.99, 'decay': 0.0}
    res = []
    for pipelined in (False, True):
        eng = Engine(0); eng.init_synthetic(7)
        tr = DataParallelTrainer(eng, world_size=1, rank=0)
        seq = data.TrainingSequence(root, dict(hp), {'image_size': 96, 'bb_info_c_size': 6})
        f = BatchFeeder(seq, 1, 0, 2)
        losses = []
        if pipelined:
            run_pipelined(eng, tr, f, [0, 1, 2], 96, hp, lambda k, loss, item: losses.append((k, float(loss.item()))))
        else:
            for k in range(3):
                losses.append((k, float(train_on_item(eng, tr, f.load(k), 96, hp).item())))
        torch.cuda.synchronize()
        f.close()
        res.append((losses, eng.params.clone()))
    assert [k for k, _ in res[1][0]] == [0, 1, 2]
    for i, ((_, a), (_, b)) in enumerate(zip(res[0][0], res[1][0])):
        assert abs(a - b) <= (1e-6 if i == 0 else 5e-3) * abs(a), (i, a, b)    # later steps: Adam amplifies the atomic-order noise
    torch.testing.assert_close(res[0][1], res[1][1], rtol=0, atol=7e-4)      # three Adam steps of lr 1e-4
