"""FaceDetector drop-in surface on the GPU: BASELINE config 1 (evaluate on 4 synthetic
UCCS-format images), detect() against the oracle chain, a tiny train() run, save/load."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _conf(root, mode, image_size=416, batch=2):
    return {'mode': mode, 'raw_data_path': root, 'test_path': root, 'output_file_path': os.path.join(root, 'solution_fd.csv'),
            'multi_gpu': False, 'num_gpus': 1, 'yolov3_base_model_load': False, 'model_loading': False,
            'hps': {'lr': 1e-4, 'beta_1': 0.99, 'beta_2': 0.99, 'decay': 0.0, 'epochs': 1, 'step': 1, 'batch_size': batch,
                    'face_conf_th': 0.5, 'nms_iou_th': 0.5, 'num_cands': 60, 'face_region_ratio_th': 0.8},
            'nn_arch': {'image_size': image_size, 'bb_info_c_size': 6}}


def test_detect_matches_oracle_chain(tmp_path, monkeypatch):
    import torch
    from face_vijnana_yolov3_amd.face_detection import FaceDetector
    from oracle import net_oracle as no
    from oracle import postproc as opp
    monkeypatch.chdir(tmp_path)
    fd = FaceDetector(_conf(str(tmp_path), 'test'))
    # random-init weights in inference mode give an arbitrary output scale: normalise the head
    # kernel to unit output std and bias it so that a healthy number of cells pass the threshold
    d = fd.model.layers[-1]
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 1, (1, 416, 416, 3))
    y0 = fd.model.predict(img.astype(np.float32))
    fd.model.params[d['w_off']:d['beta_off']] /= float(y0.std())
    fd.model.params[d['beta_off']] = 1.0; fd.model.params[d['beta_off'] + 5] = 1.0
    boxes = fd.detect(img)
    y = fd.model.predict(img.astype(np.float32))
    want = opp.detect_postproc(y, 416, 0.5, 0.5, 60)
    c = int(want['count'][0])
    assert len(boxes) == c and c > 0
    assert [(int(b.xmin), int(b.ymin), int(b.xmax), int(b.ymax)) for b in boxes] == [tuple(r) for r in want['boxes'][0, :c]]
    assert np.array_equal(np.array([b.classes[0] for b in boxes], np.float32), want['score'][0, :c])
    assert all(type(b.xmin) is np.int64 for b in boxes) and boxes[0].get_label() == 0
    # network output itself against the torch-CPU oracle (fp32 tolerance, 52 layers deep)
    yr, _ = no.forward(fd.model.params.cpu().double(), fd.model.state.cpu().double(), torch.from_numpy(img), training=False)
    y32, _ = no.forward(fd.model.params.cpu(), fd.model.state.cpu(), torch.from_numpy(img).float(), training=False)
    e_gpu = np.abs(y - yr.numpy()).max(); e_cpu = (y32.double() - yr).abs().max().item()
    assert e_gpu <= 4 * e_cpu + 1e-6 * max(1.0, yr.abs().max().item())
    # callers mutate the boxes in place (fd.py:700-710)
    boxes[0].xmin = 1.5
    assert boxes[0].xmin == 1.5


def test_config1_evaluate_plumbing(tmp_path, monkeypatch):
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.face_detection import FaceDetector
    monkeypatch.chdir(tmp_path)
    root = str(tmp_path / 'val'); os.makedirs(root)
    data.make_synthetic_uccs(root, n_images=4, seed=0, csv_name='validation.csv')
    fd = FaceDetector(_conf(root, 'evaluate'))
    d = fd.model.layers[-1]
    fd.model.params[d['w_off']:d['beta_off']] = 0          # output = bias: every cell fires
    fd.model.params[d['beta_off']] = 3.0; fd.model.params[d['beta_off'] + 5] = 3.0
    fd.model.params[d['beta_off'] + 1] = 0.5; fd.model.params[d['beta_off'] + 2] = 0.5
    fd.model.params[d['beta_off'] + 3] = 0.05; fd.model.params[d['beta_off'] + 4] = 0.05
    fd.evaluate()
    rows = [l.strip().split(',') for l in open(os.path.join(root, 'solution_fd.csv'))]
    assert len(rows) > 0 and all(len(r) == 6 for r in rows)
    names = {r[0] for r in rows}
    assert names <= {'synth_%04d.jpg' % k for k in range(4)}
    for r in rows:
        x, y, w, h, s = [float(v) for v in r[1:]]
        assert x >= 0 and y >= 0 and w >= 0 and h >= 0 and 0.5 <= s <= 1.0
    per_file = {n: sum(1 for r in rows if r[0] == n) for n in names}
    assert max(per_file.values()) <= 60
    assert len(os.listdir(os.path.join(root, 'results'))) == len(names)
    assert os.path.exists('ratios.csv')
    # test(): same rows, no drawings
    conf = _conf(root, 'test'); conf['output_file_path'] = os.path.join(root, 'solution_test.csv')
    fd.conf = conf
    fd.test()
    assert open(conf['output_file_path']).read() == open(os.path.join(root, 'solution_fd.csv')).read()


def test_tiny_train_save_load(tmp_path, monkeypatch):
    import torch
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.face_detection import FaceDetector
    monkeypatch.chdir(tmp_path)
    root = str(tmp_path / 'train'); os.makedirs(root)
    data.make_synthetic_uccs(root, n_images=5, seed=1, csv_name='training.csv')
    conf = _conf(root, 'train', image_size=96, batch=2)
    conf['hps']['epochs'] = 2
    fd = FaceDetector(conf)
    p0 = fd.model.params.clone()
    fd.train()
    assert conf['hps']['step'] == 3 and fd.model.iterations == 6
    assert not torch.equal(p0, fd.model.params) and torch.isfinite(fd.model.params).all()
    assert os.path.exists(FaceDetector.MODEL_PATH)
    from face_vijnana_yolov3_amd import hdf5_lite
    assert hdf5_lite.is_hdf5(FaceDetector.MODEL_PATH)                 # a real HDF5 file in Keras' weight layout (fd.py:630)
    _, attrs = hdf5_lite.read_hdf5(FaceDetector.MODEL_PATH)
    assert [n.decode() for n in attrs['/model_weights']['layer_names']] == ['input1', 'model_1', 'output']
    conf2 = _conf(root, 'test', image_size=96); conf2['model_loading'] = True
    fd2 = FaceDetector(conf2)
    assert torch.equal(fd2.model.params, fd.model.params) and torch.equal(fd2.model.state, fd.model.state)
    assert fd2.model.iterations == 6 and torch.equal(fd2.model.m, fd.model.m)
    x = np.random.default_rng(0).uniform(0, 1, (1, 96, 96, 3))
    assert np.array_equal(fd.model.predict(x), fd2.model.predict(x))


def test_main_reads_json_from_cwd(tmp_path, monkeypatch):
    from face_vijnana_yolov3_amd import data, face_detection
    monkeypatch.chdir(tmp_path)
    root = str(tmp_path / 'val'); os.makedirs(root)
    data.make_synthetic_uccs(root, n_images=2, seed=2, csv_name='validation.csv')
    conf = _conf(root, 'test', image_size=96)
    json.dump({'fd_conf': conf, 'fi_conf': {}}, open('face_vijnana_yolov3.json', 'w'))
    face_detection.main()
    assert os.path.exists(conf['output_file_path'])


@pytest.mark.parametrize('eval_batch', [16, 3, 1])
def test_csv_rows_match_the_reference_test_golden(tmp_path, monkeypatch, golden_dir, eval_batch):
    """a-15: the rows FaceDetector.test() writes -- letterbox geometry, detect(), the back-projection with its
    np.min/np.max clamps (fd.py:841-851), the 60-row cap and the str() formatting (fd.py:866-873) -- against
    tests/golden/test_csv.npz, minted by running the reference's own test() on the same (h, w) shapes with
    the same head output per image (make_golden.py: imread/cv2 are shape-only stand-ins there, here the
    network is replaced by the golden head the same way).  Text-identical rows, per file -- whatever the batch size
    evaluate()/test() feed the network with (the reference's loop is batch 1)."""
    import torch
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.face_detection import FaceDetector
    g = np.load(os.path.join(golden_dir, 'test_csv.npz'))
    assert str(g['numpy_version']).split('.')[0] == np.__version__.split('.')[0]     # str(np.float64) formatting era
    files = [str(f) for f in g['files']]
    hw = {f: tuple(int(v) for v in g['hw'][k]) for k, f in enumerate(files)}
    monkeypatch.chdir(tmp_path)
    root = str(tmp_path / 'imgs'); os.makedirs(root)
    for f in files:
        open(os.path.join(root, f), 'w').close()
    conf = _conf(root, 'test')
    conf['hps']['eval_batch_size'] = eval_batch
    fd = FaceDetector(conf)
    order = sorted(files)              # FaceDetector walks sorted(glob('*.jpg')) in batches of eval_batch_size, reading one batch ahead
    pos = [0]

    def loader(path):
        h, w = hw[os.path.basename(path)]
        return np.zeros((h, w, 3), np.uint8)

    def predict_device(x):
        n = int(x.shape[0])
        assert tuple(x.shape[1:]) == (416, 416, 3) and 1 <= n <= eval_batch
        names = order[pos[0]:pos[0] + n]
        pos[0] += n
        return torch.from_numpy(np.stack([g['head'][files.index(f)] for f in names])).cuda()

    monkeypatch.setattr(data, '_pil_loader', loader)
    monkeypatch.setattr(fd.model, 'predict_device', predict_device)
    fd.test()
    assert pos[0] == len(files)
    text = open(conf['output_file_path']).read().splitlines()
    for k, f in enumerate(files):
        want = str(g['rows'][k]).split('\n') if str(g['rows'][k]) else []
        got = [ln for ln in text if ln.split(',')[0] == f]
        assert len(got) == len(want), (f, len(got), len(want))
        for a, b in zip(got, want):
            # name, x, y, w, h: text-identical.  score: the float32 product of two sigmoids, where NumPy's SIMD
            # float32 exp and the kernel's correctly rounded exp may differ in the last place (as in
            # tests/test_postproc_gpu.py: scores within 2 ulp, everything else bit-exact)
            assert a.rsplit(',', 1)[0] == b.rsplit(',', 1)[0], (f, a, b)
            sa, sb = np.float32(a.rsplit(',', 1)[1]), np.float32(b.rsplit(',', 1)[1])
            assert abs(float(sa) - float(sb)) <= 2 * float(np.spacing(sb)), (f, a, b)
            assert float(a.rsplit(',', 1)[1]) == float(sa)         # printed as the float64 of a float32, like the reference
    assert sum(len(str(r).split('\n')) for r in g['rows'] if str(r)) == len(text)
    # evaluate() writes the same rows (same code path in the reference, fd.py:700-738)
    import pandas as pd
    pd.DataFrame([[0, files[0], 1, 10.0, 10.0, 20.0, 20.0]], columns=data.CSV_COLUMNS).to_csv(os.path.join(root, 'validation.csv'), index=False)
    fd.conf = dict(conf, output_file_path=os.path.join(root, 'solution_eval.csv'))
    pos[0] = 0
    fd.evaluate()
    assert open(os.path.join(root, 'solution_eval.csv')).read().splitlines() == text


def test_main_with_multi_gpu_starts_its_ranks_and_trains(tmp_path):
    """The reference's command line with fd_conf.multi_gpu / num_gpus = 2 (fd.py:358-371, 612-619): main() starts two ranks itself
    (two processes on this box's one GPU; gloo as the transport because RCCL refuses two ranks on one device), they train the
    sharded batches, rank 0 writes the model file (bit-identical replicas: tests/test_dp_rehearsal_gpu.py)."""
    import subprocess
    import sys
    from face_vijnana_yolov3_amd import data, hdf5_lite
    root = str(tmp_path / 'train'); os.makedirs(root)
    data.make_synthetic_uccs(root, n_images=6, seed=3, csv_name='training.csv')
    conf = _conf(root, 'train', image_size=96, batch=3)
    conf['multi_gpu'] = True; conf['num_gpus'] = 2
    json.dump({'fd_conf': conf, 'fi_conf': {}}, open(tmp_path / 'face_vijnana_yolov3.json', 'w'))
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(FV_DIST_BACKEND='gloo', FV_DEVICE='0', PYTHONPATH=repo + os.pathsep + env.get('PYTHONPATH', ''))
    r = subprocess.run([sys.executable, '-m', 'face_vijnana_yolov3_amd.face_detection'], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'starting 2 ranks' in r.stderr
    assert r.stdout.count('Save the model.') == 1 and 'Epoch 1/1' in r.stdout
    assert hdf5_lite.is_hdf5(str(tmp_path / 'face_detector.h5'))
    ds, _ = hdf5_lite.read_hdf5(str(tmp_path / 'face_detector.h5'))
    assert int(np.asarray(ds['/fv/iterations']).reshape(-1)[0]) == 2          # 6 images, merged batch 3 (2 + 1 per rank), one epoch
    assert all(np.isfinite(v).all() for k, v in ds.items() if v.dtype.kind == 'f')


@pytest.mark.parametrize('mode', ['evaluate', 'test'])
def test_rows_do_not_depend_on_the_eval_batch_size_with_real_files(tmp_path, monkeypatch, mode):
    """Seven real JPEGs through evaluate() (Pillow decode, pixels kept for drawing) and test() (host / device JPEG split): more
    batches than the three reused pinned buffers, a short last batch, the two-deep device queue -- the csv holds the same rows at
    eval_batch_size 1, 2 and 32 (text-identical for a fixed head output: test_csv_rows_match_the_reference_test_golden)."""
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.face_detection import FaceDetector
    monkeypatch.chdir(tmp_path)
    root = str(tmp_path / 'val'); os.makedirs(root)
    data.make_synthetic_uccs(root, n_images=7, seed=4, csv_name='validation.csv')
    texts = {}
    fd = None
    for bs in (1, 2, 32):
        conf = _conf(root, mode)
        conf['hps']['eval_batch_size'] = bs
        conf['output_file_path'] = os.path.join(root, 'solution_%d.csv' % bs)
        if fd is None:
            fd = FaceDetector(conf)
            d = fd.model.layers[-1]                                # a head of unit output scale that fires on some cells
            y0 = fd.model.predict(np.random.default_rng(0).uniform(0, 1, (1, 416, 416, 3)).astype(np.float32))
            fd.model.params[d['w_off']:d['beta_off']] /= float(y0.std())
            fd.model.params[d['beta_off']] = 1.0; fd.model.params[d['beta_off'] + 5] = 1.0
        fd.conf = conf; fd.hps = conf['hps']
        getattr(fd, mode)()
        texts[bs] = open(conf['output_file_path']).read()
    # The network itself is not bit-identical across batch sizes (the K-split / tail plans of the conv launches, i.e. the fp32
    # summation order, depend on the tile count), so: same files, same number of rows per file, same order, boxes equal up
    # to one truncation step of the back-projection, scores to float32 rounding of the head.
    rows = {bs: [ln.split(',') for ln in texts[bs].splitlines()] for bs in texts}
    assert rows[1] and len({r[0] for r in rows[1]}) >= 2
    for bs in (2, 32):
        assert [r[0] for r in rows[bs]] == [r[0] for r in rows[1]], bs
        a = np.array([[float(v) for v in r[1:]] for r in rows[bs]]); b = np.array([[float(v) for v in r[1:]] for r in rows[1]])
        assert np.abs(a[:, :4] - b[:, :4]).max() <= 1e-6 * max(1.0, np.abs(b[:, :4]).max()), bs
        assert np.abs(a[:, 4] - b[:, 4]).max() <= 2e-6, bs


@pytest.mark.parametrize('head', ['single', 'three_scale'])
def test_yolov3_base_property(tmp_path, monkeypatch, head):
    """FaceDetector.YOLOV3Base (fd.py:384-600; SURVEY 1, L3 API): the Darknet-53 base as a model of its own -- predict gives the
    add_23 output the head reads; base.save writes the yolov3_base.h5 layout (fd.py:598), which a detector configured with
    yolov3_base_model_load then loads (fd.py:393-396)."""
    import torch
    from face_vijnana_yolov3_amd.face_detection import FaceDetector
    from face_vijnana_yolov3_amd.engine import Engine
    monkeypatch.chdir(tmp_path)
    conf = _conf(str(tmp_path), 'test', image_size=96)
    conf['nn_arch']['head'] = head
    fd = FaceDetector(conf)
    base = fd.YOLOV3Base
    assert base.trainable is True
    x = np.random.default_rng(1).uniform(0, 1, (2, 96, 96, 3)).astype(np.float32)
    f = base.predict(x)
    assert f.shape == (2, 3, 3, 1024) and f.dtype == np.float32 and np.isfinite(f).all() and f.std() > 0
    if head == 'single':
        # the head conv applied to these features is the detector's own output
        feat, y = fd.model.predict_base_device(torch.from_numpy(x), with_head=True)
        assert np.array_equal(feat.cpu().numpy(), f) and np.array_equal(y.cpu().numpy(), fd.model.predict(x))
    base.save('yolov3_base.h5')
    conf2 = _conf(str(tmp_path), 'test', image_size=96)
    conf2['yolov3_base_model_load'] = True
    fd2 = FaceDetector(conf2)
    assert np.array_equal(fd2.YOLOV3Base.predict(x), f)
