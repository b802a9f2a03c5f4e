#!/usr/bin/env python3
"""Mint golden vectors by RUNNING the reference's own functions (build container only).

This script imports /root/reference/src/space/{yolov3_detect,face_detection}.py with
empty stand-in modules for the third-party packages that are not installed here
(keras, cv2, skimage) -- none of the functions exercised below call into them,
except the GT encoder, whose imread/resize/copyMakeBorder are shape-only stand-ins
(pixel content is NOT pinned, see DESIGN.md "parity unpinned" list).

Outputs (data only -- inputs and expected outputs; no reference source text):
  tests/golden/detect_cases.npz    FaceDetector.detect  (fd.py:885-949, yd.py:446-458)
  tests/golden/iou_cases.npz       bbox_iou             (yd.py:165-194)
  tests/golden/gt_encoder.npz      TrainingSequence.__getitem__ GT tensors (fd.py:98-310)
  tests/golden/weight_reader.npz   WeightReader         (yd.py:67-124)
  tests/golden/decode_netout.npz   decode_netout/do_nms/correct_yolo_boxes (yd.py:335-444)
  tests/golden/decode_netout_coco80.npz  the same on 80 classes and a small image (more int() boundary cases)
  tests/golden/test_csv.npz        FaceDetector.test() csv rows: letterbox geometry, detect,
                                   back-projection and str() formatting (fd.py:783-883)

The reference does not exist on the GPU box; nothing at test time imports this file.
Run:  python tests/golden/make_golden.py
"""
import io
import os
import struct
import sys
import tempfile
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/src/space'


# --------------------------------------------------------------------------- stubs
class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, n):
        return _Any()


def _install_stubs():
    def stub(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    for n in ['keras', 'keras.layers', 'keras.layers.merge', 'keras.models', 'keras.utils',
              'keras.utils.data_utils', 'keras.optimizers', 'keras.backend', 'skimage',
              'skimage.io', 'skimage.transform', 'skimage.draw', 'cv2']:
        stub(n)
    for n in ['Conv2D', 'Input', 'BatchNormalization', 'LeakyReLU', 'ZeroPadding2D',
              'UpSampling2D', 'Lambda', 'Concatenate']:
        setattr(sys.modules['keras.layers'], n, _Any)
    sys.modules['keras.layers.merge'].add = _Any
    sys.modules['keras.layers.merge'].concatenate = _Any
    sys.modules['keras.models'].Model = _Any
    sys.modules['keras.models'].load_model = _Any
    sys.modules['keras.utils'].multi_gpu_model = _Any
    sys.modules['keras'].optimizers = _Any()
    sys.modules['keras'].backend = _Any()

    class Sequence:  # keras.utils.data_utils.Sequence must be a real base class
        pass

    sys.modules['keras.utils.data_utils'].Sequence = Sequence
    sys.modules['skimage.io'].imread = None
    sys.modules['skimage.io'].imsave = None
    sys.modules['skimage.transform'].resize = None
    sys.modules['skimage.draw'].polygon_perimeter = None
    sys.modules['skimage.draw'].set_color = None


_install_stubs()
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import yolov3_detect as yd  # noqa: E402
import face_detection as fd  # noqa: E402


# --------------------------------------------------------------------------- detect
def synth_head(rng, n, grid=13, obj_sigma=2.0):
    """Config-4 style head outputs."""
    y = np.zeros((n, grid, grid, 6), np.float32)
    y[..., 0] = rng.normal(0, obj_sigma, (n, grid, grid))
    y[..., 5] = rng.normal(0, obj_sigma, (n, grid, grid))
    y[..., 1:3] = rng.uniform(0, 1, (n, grid, grid, 2))
    y[..., 3:5] = rng.uniform(0, 0.3, (n, grid, grid, 2))
    return y


class _FakeModel:
    def __init__(self, y):
        self.y = y

    def predict(self, x):
        return self.y.copy()


def run_detect(y1, conf_th, iou_th, num_cands, image_size=416):
    """Drive the real FaceDetector.detect on one (1,13,13,6) float32 head output."""
    det = fd.FaceDetector.__new__(fd.FaceDetector)
    det.hps = {'face_conf_th': conf_th, 'nms_iou_th': iou_th, 'num_cands': num_cands}
    det.nn_arch = {'image_size': image_size}
    det.cell_image_size = image_size // fd.FaceDetector.CELL_SIZE
    det.model = _FakeModel(y1)

    created = []
    real_bb = yd.BoundBox

    class RecBB(real_bb):
        def __init__(self, *a, **k):
            real_bb.__init__(self, *a, **k)
            self.seq = len(created)
            created.append(self)

    fd.BoundBox = RecBB
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            out = det.detect(None)
    finally:
        fd.BoundBox = real_bb

    # seq -> cell map: values computed with the reference's own _sigmoid (fd.py:904-905)
    fc = np.squeeze(y1.copy())
    fc[..., 0] = yd._sigmoid(fc[..., 0])
    fc[..., -1] = fc[..., 0] * yd._sigmoid(fc[..., -1])
    cells = [(i, j) for i in range(fc.shape[0]) for j in range(fc.shape[1])
             if fc[i, j, 0] > 0. and fc[i, j, -1] >= conf_th]
    assert len(cells) == len(created), (len(cells), len(created))
    for b, (i, j) in zip(created, cells):
        assert b.objness == fc[i, j, 0]
    cand_scores = np.array([fc[i, j, -1] for (i, j) in cells], np.float32)
    res = np.zeros((len(out), 7), np.float64)
    for k, b in enumerate(out):
        i, j = cells[b.seq]
        res[k] = [b.xmin, b.ymin, b.xmax, b.ymax, i * fc.shape[1] + j, b.objness, b.get_score()]
    return res, cand_scores, len(created)


def edge_cases():
    cases = []
    z = lambda: np.zeros((1, 13, 13, 6), np.float32)
    # E0: nothing above threshold
    y = z(); y[..., 0] = -5; y[..., 5] = -5
    cases.append(('none', y, 0.5, 0.5, 60))
    # E1: exactly at threshold: sigma(40)=1.0f exactly, sigma(0)=0.5 -> score 0.5 passes '>='
    y = z(); y[..., 0] = -9; y[..., 5] = -9
    y[0, 3, 4] = [40, .5, .5, .1, .1, 0]; y[0, 7, 7] = [40, .25, .75, .2, .1, 0.001]
    cases.append(('at_threshold', y, 0.5, 0.5, 60))
    # E2: >60 survivors, no overlaps (20 px boxes in 32 px cells): lowest 60 kept, ascending
    rng = np.random.default_rng(5)
    y = z(); y[..., 0] = rng.uniform(3, 6, (1, 13, 13)); y[..., 5] = rng.uniform(1, 5, (1, 13, 13))
    y[..., 1:3] = 0.5; y[..., 3:5] = 0.05
    cases.append(('more_than_60', y, 0.5, 0.5, 60))
    # E3: heavy overlaps: big boxes everywhere (chain suppression)
    rng = np.random.default_rng(6)
    y = z(); y[..., 0] = rng.uniform(2, 6, (1, 13, 13)); y[..., 5] = rng.uniform(0.5, 5, (1, 13, 13))
    y[..., 1:3] = rng.uniform(0, 1, (1, 13, 13, 2)); y[..., 3:5] = rng.uniform(0.3, 0.6, (1, 13, 13, 2))
    cases.append(('heavy_overlap', y, 0.5, 0.5, 60))
    # E4: zero-area boxes (bw=bh=0 -> nan IoU between them, no suppression) mixed with normal
    rng = np.random.default_rng(7)
    y = z(); y[..., 0] = rng.uniform(2, 6, (1, 13, 13)); y[..., 5] = rng.uniform(0.5, 5, (1, 13, 13))
    y[..., 1:3] = rng.uniform(0, 1, (1, 13, 13, 2)); y[..., 3:5] = rng.uniform(0.0, 0.4, (1, 13, 13, 2))
    y[0, ::2, ::2, 3:5] = 0.0
    y[0, 1::4, :, 3] = 0.0
    cases.append(('zero_area', y, 0.5, 0.5, 60))
    # E5: out-of-range regressions: negative bx/by/bw/bh (clip 0), bx>1 (min cs-1), bw>1 (clamp S)
    rng = np.random.default_rng(8)
    y = z(); y[..., 0] = rng.uniform(1, 6, (1, 13, 13)); y[..., 5] = rng.uniform(0.5, 5, (1, 13, 13))
    y[..., 1:3] = rng.uniform(-0.5, 1.8, (1, 13, 13, 2)); y[..., 3:5] = rng.uniform(-0.2, 1.6, (1, 13, 13, 2))
    cases.append(('out_of_range', y, 0.5, 0.5, 60))
    # E6/E7: other thresholds (exactly representable in float32) and small num_cands
    rng = np.random.default_rng(9)
    cases.append(('th_025', synth_head(rng, 1), 0.25, 0.25, 10))
    cases.append(('th_075', synth_head(rng, 1, obj_sigma=4.0), 0.75, 0.75, 5))
    # E8: iou threshold 0 -> 'iou >= 0' suppresses every later box that is not nan
    cases.append(('iou_zero', synth_head(rng, 1), 0.5, 0.0, 60))
    # E9: iou threshold 1.0 (only exact coincidence suppresses)
    cases.append(('iou_one', synth_head(rng, 1), 0.5, 1.0, 60))
    # E10: a single candidate
    y = z(); y[..., 0] = -9; y[0, 12, 12] = [3, .9, .9, .5, .5, 3]
    cases.append(('single', y, 0.5, 0.5, 60))
    return cases


def mint_detect():
    rng = np.random.default_rng(99)
    heads, meta, outs, cand_scores = [], [], [], []
    names = []
    # random config-4 style frames (tie-free)
    n_rand = 0
    while n_rand < 96:
        y = synth_head(rng, 1)
        res, cs, ncand = run_detect(y, 0.5, 0.5, 60)
        if len(np.unique(cs)) != len(cs):
            continue  # exact ties are outside the parity contract
        heads.append(y[0]); meta.append([0.5, 0.5, 60, ncand]); outs.append(res); cand_scores.append(cs)
        names.append('rand%d' % n_rand)
        n_rand += 1
    for name, y, cth, ith, nc in edge_cases():
        res, cs, ncand = run_detect(y, cth, ith, nc)
        assert len(np.unique(cs)) == len(cs), name
        heads.append(y[0]); meta.append([cth, ith, nc, ncand]); outs.append(res); cand_scores.append(cs)
        names.append(name)
    n = len(heads)
    maxo = max(len(o) for o in outs)
    out_arr = np.full((n, max(maxo, 1), 7), -1.0, np.float64)
    out_cnt = np.zeros(n, np.int32)
    for k, o in enumerate(outs):
        out_cnt[k] = len(o)
        if len(o):
            out_arr[k, :len(o)] = o
    np.savez_compressed(os.path.join(HERE, 'detect_cases.npz'),
                        head=np.stack(heads), meta=np.array(meta, np.float64),
                        out=out_arr, out_count=out_cnt, names=np.array(names))
    print('detect_cases:', n, 'cases; max out', maxo, '; counts', out_cnt[-12:])


# --------------------------------------------------------------------------- iou
def mint_iou():
    rng = np.random.default_rng(3)
    n = 4000
    a = rng.integers(0, 416, (n, 4)); b = rng.integers(0, 416, (n, 4))
    for arr in (a, b):
        x0 = np.minimum(arr[:, 0], arr[:, 2]); x1 = np.maximum(arr[:, 0], arr[:, 2])
        y0 = np.minimum(arr[:, 1], arr[:, 3]); y1 = np.maximum(arr[:, 1], arr[:, 3])
        arr[:, 0], arr[:, 1], arr[:, 2], arr[:, 3] = x0, y0, x1, y1
    # force degenerate / touching / nested / identical cases
    a[:50, 2] = a[:50, 0]; b[:50, 2] = b[:50, 0]        # zero width both  -> 0/0 = nan
    a[50:100, 3] = a[50:100, 1]                          # zero height in a only
    b[100:150] = a[100:150]                              # identical
    b[150:200, 0] = a[150:200, 2]                        # touching edge x3 == x2
    b[150:200, 2] = np.maximum(b[150:200, 2], b[150:200, 0])
    out = np.zeros(n, np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for k in range(n):
            b1 = yd.BoundBox(*[np.int64(v) for v in a[k]]); b2 = yd.BoundBox(*[np.int64(v) for v in b[k]])
            out[k] = yd.bbox_iou(b1, b2)
    np.savez_compressed(os.path.join(HERE, 'iou_cases.npz'), a=a.astype(np.int32), b=b.astype(np.int32), iou=out)
    print('iou_cases:', n, 'nan', int(np.isnan(out).sum()), 'inf', int(np.isinf(out).sum()))


# --------------------------------------------------------------------------- GT encoder
def mint_gt():
    import pandas as pd
    rng = np.random.default_rng(11)
    sizes = {}  # file -> (h, w)
    rows = []
    files = ['img_%03d.jpg' % k for k in range(11)]
    hw = [(300, 500), (500, 300), (416, 416), (480, 640), (640, 480), (601, 1000), (1000, 601),
          (333, 777), (777, 333), (1080, 1920), (123, 124)]
    fid = 0
    for f, (h, w) in zip(files, hw):
        sizes[f] = (h, w)
        nf = int(rng.integers(1, 7))
        for _ in range(nf):
            fw = float(rng.uniform(8, w / 3)); fh = float(rng.uniform(8, h / 3))
            fx = float(rng.uniform(1, w - fw - 1)); fy = float(rng.uniform(1, h - fh - 1))
            rows.append([fid, f, int(rng.integers(-1, 50)), round(fx, 2), round(fy, 2), round(fw, 2), round(fh, 2)])
            fid += 1
    # hand-made rows: the SURVEY example, a skipped row (FACE_X = 0), same-cell overwrite
    rows.append([fid, files[0], -1, 100.5, 50.2, 40, 60]); fid += 1
    rows.append([fid, files[0], 3, 0, 50.2, 40, 60]); fid += 1           # skipped (X not > 0)
    rows.append([fid, files[2], 4, 200.0, 200.0, 20, 20]); fid += 1
    rows.append([fid, files[2], 5, 203.0, 201.0, 30, 26]); fid += 1      # same cell -> overwrite
    rows.append([fid, files[3], 6, 10.0, 10.0, -5, 20]); fid += 1        # skipped (W not > 0)
    df = pd.DataFrame(rows, columns=['FACE_ID', 'FILE', 'SUBJECT_ID', 'FACE_X', 'FACE_Y', 'FACE_WIDTH', 'FACE_HEIGHT'])

    def imread(path):
        h, w = sizes[os.path.basename(path)]
        return np.zeros((h, w, 3), np.uint8)

    class CV:
        INTER_CUBIC = 2
        BORDER_CONSTANT = 0

        @staticmethod
        def resize(img, dsize, interpolation=None):
            return np.zeros((dsize[1], dsize[0], 3), np.float64)

        @staticmethod
        def copyMakeBorder(img, t, b, l, r, mode, value=None):
            return np.pad(img, ((t, b), (l, r), (0, 0)))

    fd.imread = imread
    fd.cv = CV
    batch = 4
    with tempfile.TemporaryDirectory() as d:
        df.to_csv(os.path.join(d, 'training.csv'), index=False)
        hps = {'batch_size': batch, 'step': 1}
        seq = fd.FaceDetector.TrainingSequence(d, hps, {'image_size': 416, 'bb_info_c_size': 6}, 13, 32)
        gts, shapes, counts = [], [], []
        for k in range(len(seq)):
            x, y = seq[k]
            gts.append(y['output']); shapes.append(x['input1'].shape); counts.append(y['output'].shape[0])
        file_names = list(seq.file_names)
    gt = np.concatenate(gts, 0)
    csv_buf = io.StringIO(); df.to_csv(csv_buf, index=False)
    np.savez_compressed(os.path.join(HERE, 'gt_encoder.npz'),
                        csv=np.array(csv_buf.getvalue()), files=np.array(file_names),
                        hw=np.array([sizes[f] for f in file_names], np.int32),
                        gt=gt, batch_size=np.int32(batch), step=np.int32(hps['step']),
                        batch_counts=np.array(counts, np.int32),
                        image_shapes=np.array(shapes, np.int32))
    print('gt_encoder:', gt.shape, 'step', hps['step'], 'counts', counts, 'nonzero cells', int((gt[..., 0] > 0).sum()))


# --------------------------------------------------------------------------- WeightReader
def mint_weight_reader():
    # tiny duck-typed model: conv_0 (bn), conv_1 (bn), conv_3 (bn), conv_81 (bias, no bn); others missing
    spec = {0: ((3, 3, 3, 4), True), 1: ((3, 3, 4, 8), True), 3: ((1, 1, 8, 6), True), 81: ((1, 1, 6, 5), False)}

    class Layer:
        def __init__(self, shapes):
            self.w = [np.zeros(s, np.float32) for s in shapes]
            self.set = None

        def get_weights(self):
            return self.w

        def set_weights(self, ws):
            self.set = [np.array(w) for w in ws]

    class Mdl:
        def __init__(self):
            self.layers = {}
            for i, (shape, bn) in spec.items():
                self.layers['conv_%d' % i] = Layer([shape] if bn else [shape, (shape[3],)])
                if bn:
                    self.layers['bnorm_%d' % i] = Layer([(shape[3],)] * 4)

        def get_layer(self, name):
            if name not in self.layers:
                raise ValueError(name)
            return self.layers[name]

    total = sum(int(np.prod(s)) + (4 * s[3] if bn else s[3]) for s, bn in spec.values())
    rng = np.random.default_rng(21)
    payload = rng.standard_normal(total + 7).astype(np.float32)  # + trailing unread floats
    results = {}
    for tag, header in (('v2', struct.pack('iii', 0, 2, 0) + struct.pack('q', 32013312)),
                        ('v1', struct.pack('iii', 0, 1, 0) + struct.pack('i', 12345))):
        with tempfile.NamedTemporaryFile(suffix='.weights', delete=False) as f:
            f.write(header); f.write(payload.tobytes()); path = f.name
        import contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            m = Mdl(); wr = yd.WeightReader(path); wr.load_weights(m)
        os.unlink(path)
        results[tag + '_file'] = np.frombuffer(header + payload.tobytes(), np.uint8)
        results[tag + '_offset'] = np.int64(wr.offset)
        for name, lyr in m.layers.items():
            for k, w in enumerate(lyr.set):
                results['%s_%s_%d' % (tag, name, k)] = w
    np.savez_compressed(os.path.join(HERE, 'weight_reader.npz'), **results)
    print('weight_reader: total floats', total, 'offsets', results['v2_offset'], results['v1_offset'])


# --------------------------------------------------------------------------- decode_netout (secondary)
def mint_decode_netout():
    rng = np.random.default_rng(31)
    anchors = [[116, 90, 156, 198, 373, 326], [30, 61, 62, 45, 59, 119], [10, 13, 16, 30, 33, 23]]
    ncls = 4
    net_h = net_w = 416
    image_h, image_w = 1440, 1920  # large: int() corners never collapse to zero area (python-int 0/0 raises in yd.py:194)
    outs = {}
    boxes = []
    for s, g in enumerate((13, 26, 52)):
        no = rng.normal(0, 1.5, (g, g, 3 * (5 + ncls))).astype(np.float32)
        no4 = no.reshape(g, g, 3, -1); no4[..., 4] -= 2.0; no4[..., 2:4] *= 0.3
        outs['netout_%d' % s] = no.copy()
        boxes += yd.decode_netout(no.copy(), anchors[s], s, 0.5, net_h, net_w)
    pre = np.array([[b.xmin, b.ymin, b.xmax, b.ymax, b.objness] + list(b.classes) for b in boxes], np.float64)
    yd.correct_yolo_boxes(boxes, image_h, image_w, net_h, net_w)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        yd.do_nms(boxes, 0.5)
    post = np.array([[b.xmin, b.ymin, b.xmax, b.ymax, b.objness] + list(b.classes) for b in boxes], np.float64)
    np.savez_compressed(os.path.join(HERE, 'decode_netout.npz'), pre=pre, post=post,
                        anchors=np.array(anchors, np.int32), image_hw=np.array([image_h, image_w], np.int32), **outs)
    print('decode_netout:', pre.shape, 'suppressed entries', int(((pre[:, 5:] != 0) & (post[:, 5:] == 0)).sum()))


# --------------------------------------------------------------------------- test() csv rows (a-15)
def mint_test_csv():
    """Run the reference's FaceDetector.test() (fd.py:783-883) end to end on synthetic (h, w) shapes with
    a chosen head output per image: imread / cv2 are shape-only stand-ins (pixels do not matter: the
    fake predict ignores them), everything else -- pad computation, detect(), box back-projection with its
    np.min/np.max clamps, the 60-row cap and the str() formatting of the rows -- is the reference's code."""
    import contextlib
    rng = np.random.default_rng(41)
    hw = [(480, 640), (640, 480), (416, 416), (601, 1000), (1000, 601), (333, 777), (123, 124), (1080, 1920), (97, 31), (50, 400)]
    files = ['t_%03d.jpg' % k for k in range(len(hw))]
    sizes = dict(zip(files, hw))
    heads = {f: synth_head(rng, 1, obj_sigma=2.0 + 0.3 * k) for k, f in enumerate(files)}
    heads[files[2]][..., 0] = -9.0            # an image with no detections: no rows at all
    heads[files[8]][..., 3:5] *= 4.0          # big boxes on the extreme portrait: clamps at w / h
    heads[files[9]][..., 1:3] = rng.uniform(-0.3, 1.3, (1, 13, 13, 2))
    seen = []

    def imread(path):
        seen.append(os.path.basename(path))
        h, w = sizes[os.path.basename(path)]
        return np.zeros((h, w, 3), np.uint8)

    class CV:
        INTER_CUBIC = 2
        BORDER_CONSTANT = 0

        @staticmethod
        def resize(img, dsize, interpolation=None):
            return np.zeros((dsize[1], dsize[0], 3), np.float64)

        @staticmethod
        def copyMakeBorder(img, t, b, l, r, mode, value=None):
            return np.pad(img, ((t, b), (l, r), (0, 0)))

    class Model:
        def predict(self, x):
            assert x.shape == (1, 416, 416, 3), x.shape      # the letterbox produced S x S
            return heads[seen[-1]].copy()

    fd.imread = imread
    fd.cv = CV
    with tempfile.TemporaryDirectory() as d:
        for f in files:
            open(os.path.join(d, f), 'w').close()
        det = fd.FaceDetector.__new__(fd.FaceDetector)
        det.conf = {'test_path': d, 'output_file_path': os.path.join(d, 'solution.csv')}
        det.hps = {'face_conf_th': 0.5, 'nms_iou_th': 0.5, 'num_cands': 60}
        det.nn_arch = {'image_size': 416}
        det.cell_image_size = 416 // fd.FaceDetector.CELL_SIZE
        det.model = Model()
        with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
            warnings.simplefilter('ignore')
            det.test()
        text = open(det.conf['output_file_path']).read()
    rows = {f: [ln for ln in text.splitlines() if ln.split(',')[0] == f] for f in files}
    assert sum(len(v) for v in rows.values()) == len(text.splitlines()) and not rows[files[2]]
    np.savez_compressed(os.path.join(HERE, 'test_csv.npz'), files=np.array(files), hw=np.array(hw, np.int32),
                        head=np.stack([heads[f][0] for f in files]),
                        rows=np.array(['\n'.join(rows[f]) for f in files]), numpy_version=np.array(np.__version__))
    print('test_csv:', len(files), 'images', [len(rows[f]) for f in files], 'rows; e.g.', rows[files[0]][0])


def mint_decode_netout_coco80():
    """Second decode_netout fixture: 80 classes (the COCO demo's real width) and a SMALL image, where the
    int() truncation of correct_yolo_boxes lands on many more integer boundaries.  The head outputs are a
    quiet background (objectness logit -12: below any threshold) plus a few hundred "hot" cells, stored
    sparsely (cell index + 255 values) to keep the fixture small.  Pairs whose corrected boxes have zero
    area would raise ZeroDivisionError in bbox_iou (yd.py:194, python ints): checked that none collapses."""
    rng = np.random.default_rng(57)
    anchors = [[116, 90, 156, 198, 373, 326], [30, 61, 62, 45, 59, 119], [10, 13, 16, 30, 33, 23]]
    ncls = 80
    net_h = net_w = 416
    image_h, image_w = 375, 500
    sparse = {}
    boxes = []
    for s, (g, nhot) in enumerate(((13, 60), (26, 110), (52, 160))):
        no = np.zeros((g, g, 3 * (5 + ncls)), np.float32)
        no.reshape(g, g, 3, -1)[..., 4] = -12.0
        # hot cells in clusters (neighbouring cells with boxes larger than a cell overlap -> suppression)
        centres = rng.integers(1, g - 1, (nhot // 5, 2))
        cells = np.unique(np.clip(np.repeat(centres, 5, 0) + rng.integers(-1, 2, (nhot // 5 * 5, 2)), 0, g - 1), axis=0)
        vals = rng.normal(0, 1.0, (len(cells), 3, 5 + ncls)).astype(np.float32)
        vals[..., 4] += 1.0; vals[..., 2:4] = vals[..., 2:4] * 0.25 + 0.5
        vals[..., 5:] -= 2.0
        fav = rng.integers(0, ncls, len(cells))                      # clusters agree on a few classes
        for k in range(len(cells)):
            vals[k, :, 5 + fav[k] % 7] += 4.0
        vals = vals.reshape(len(cells), -1)
        no[cells[:, 0], cells[:, 1]] = vals
        sparse['cells_%d' % s] = cells.astype(np.int32); sparse['vals_%d' % s] = vals
        boxes += yd.decode_netout(no.copy(), anchors[s], s, 0.5, net_h, net_w)
    pre = np.array([[b.xmin, b.ymin, b.xmax, b.ymax, b.objness] + list(b.classes) for b in boxes], np.float64)
    yd.correct_yolo_boxes(boxes, image_h, image_w, net_h, net_w)
    assert all(b.xmax > b.xmin and b.ymax > b.ymin for b in boxes), 'zero-area box: redraw'
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        yd.do_nms(boxes, 0.5)
    post = np.array([[b.xmin, b.ymin, b.xmax, b.ymax, b.objness] + list(b.classes) for b in boxes], np.float64)
    np.savez_compressed(os.path.join(HERE, 'decode_netout_coco80.npz'), pre=pre, post=post,
                        anchors=np.array(anchors, np.int32), image_hw=np.array([image_h, image_w], np.int32),
                        background_obj=np.float32(-12.0), **sparse)
    print('decode_netout_coco80:', pre.shape, 'suppressed entries', int(((pre[:, 5:] != 0) & (post[:, 5:] == 0)).sum()))


if __name__ == '__main__':
    which = sys.argv[1:] or ['detect', 'iou', 'gt', 'weight_reader', 'decode_netout', 'test_csv', 'decode_netout_coco80']
    for name in which:
        {'detect': mint_detect, 'iou': mint_iou, 'gt': mint_gt, 'weight_reader': mint_weight_reader,
         'decode_netout': mint_decode_netout, 'test_csv': mint_test_csv, 'decode_netout_coco80': mint_decode_netout_coco80}[name]()
