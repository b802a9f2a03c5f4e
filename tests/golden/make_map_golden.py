#!/opt/conda/bin/python3.9
"""Mint golden vectors for cal_mAP_fd by RUNNING the reference's own function (evaluate.py:27-127).  Build container only:

    /opt/conda/bin/python3.9 tests/golden/make_map_golden.py      # the interpreter that has h5py, which evaluate.py imports

As committed, the reference function raises at evaluate.py:31 under every pandas release: `sol_df.iat[:, 6] = -1.0` (and the
same at :36) hands a SLICE to `.iat`, which only takes integer positions.  What the two lines mean is not in doubt -- "set the
new IoU column to -1" --, so for exactly that key shape (a full slice and an integer column) `.iat.__setitem__` is shimmed to
the `.iloc` assignment; every other `.iat` use of the function (`rel_sol_df.iat[j, -1] = iou`, evaluate.py:89) goes through
pandas untouched.  Nothing else of the reference is altered; keras / cv2 / skimage (imported by yolov3_detect, unused here) are
empty stand-in modules as in make_golden.py.

The inputs avoid what the committed text leaves undefined: the first ground-truth image (groupby order) has an overlapping
detection (otherwise `res_df` is never bound, evaluate.py:97-100), and there are no ties in IoU or confidence (`sort_values`
is not stable).  Output: tests/golden/cal_map_fd.npz -- the two csv texts per case and (ps, rs, mAP) at several thresholds.
Data only; no reference source text."""
import io
import os
import sys
import tempfile
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/src/space'


class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, n):
        return _Any()


def _install_stubs():
    for n in ['keras', 'keras.layers', 'keras.layers.merge', 'keras.models', 'keras.utils', 'keras.utils.data_utils',
              'keras.optimizers', 'keras.backend', 'skimage', 'skimage.io', 'skimage.transform', 'skimage.draw', 'cv2']:
        sys.modules[n] = types.ModuleType(n)
    for n in ['Conv2D', 'Input', 'BatchNormalization', 'LeakyReLU', 'ZeroPadding2D', 'UpSampling2D', 'Lambda', 'Concatenate']:
        setattr(sys.modules['keras.layers'], n, _Any)
    sys.modules['keras.layers.merge'].add = _Any
    sys.modules['keras.layers.merge'].concatenate = _Any
    sys.modules['keras.models'].Model = _Any
    sys.modules['keras.models'].load_model = _Any
    sys.modules['keras.utils'].multi_gpu_model = _Any
    sys.modules['keras'].optimizers = _Any()
    sys.modules['keras'].backend = _Any()
    sys.modules['keras.utils.data_utils'].Sequence = type('Sequence', (), {})
    for n in ('imread', 'imsave'):
        setattr(sys.modules['skimage.io'], n, None)
    sys.modules['skimage.transform'].resize = None
    sys.modules['skimage.draw'].polygon_perimeter = None
    sys.modules['skimage.draw'].set_color = None


def _shim_iat():
    from pandas.core.indexing import _iAtIndexer
    orig = _iAtIndexer.__setitem__
    used = []

    def setitem(self, key, value):
        if isinstance(key, tuple) and len(key) == 2 and isinstance(key[0], slice) and key[0] == slice(None) and isinstance(key[1], int):
            used.append(key[1])
            self.obj.iloc[:, key[1]] = value          # evaluate.py:31, 36
            return
        return orig(self, key, value)

    _iAtIndexer.__setitem__ = setitem
    return used


def synth(seed, n_img, first_has_match=True):
    """-> (gt csv text, sol csv text).  Faces with jittered detections, false positives, images without detections, an image
    whose only detection overlaps nothing, a detection on an image that is not in the ground truth."""
    rng = np.random.default_rng(seed)
    gt = ['FACE_ID,FILE,SUBJECT_ID,FACE_X,FACE_Y,FACE_WIDTH,FACE_HEIGHT']
    sol = []
    fid = 0
    for k in range(n_img):
        name = 'img_%03d.jpg' % k
        kind = 'match' if k == 0 and first_has_match else rng.choice(['match', 'match', 'match', 'none', 'far'])
        for f in range(int(rng.integers(1, 5))):
            x, y = rng.uniform(1, 700, 2); w, h = rng.uniform(24, 140, 2)
            gt.append('%d,%s,%d,%.1f,%.1f,%.1f,%.1f' % (fid, name, int(rng.integers(1, 50)), x, y, w, h)); fid += 1
            if kind == 'match' and (rng.random() < 0.85 or (k == 0 and f == 0)):
                j = rng.normal(0, 0.12, 4) * np.array([w, h, w, h])
                sol.append('%s,%r,%r,%r,%r,%r' % (name, float(x + j[0]), float(y + j[1]), float(max(w + j[2], 4)), float(max(h + j[3], 4)), float(rng.uniform(0.3, 1.0))))
        if kind == 'match':
            for _ in range(int(rng.integers(0, 3))):       # false positives, possibly overlapping something
                sol.append('%s,%r,%r,%r,%r,%r' % (name, float(rng.uniform(1, 800)), float(rng.uniform(1, 800)), float(rng.uniform(10, 90)), float(rng.uniform(10, 90)), float(rng.uniform(0.05, 0.9))))
        elif kind == 'far':
            sol.append('%s,%r,%r,%r,%r,%r' % (name, 5000.0 + float(rng.uniform(0, 9)), 5000.0, 12.0, 12.0, float(rng.uniform(0.5, 0.99))))
    sol.append('zz_not_in_gt.jpg,%r,%r,%r,%r,%r' % (1.0, 1.0, 10.0, 10.0, 0.987654))
    return '\n'.join(gt) + '\n', '\n'.join(sol) + '\n'


def main():
    _install_stubs()
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    warnings.simplefilter('ignore')
    used = _shim_iat()
    import evaluate as ref                                   # the reference's module
    ref.DEBUG = False
    out = {}
    ths = [0.3, 0.5, 0.65, 0.8, 0.95]
    cases = [(0, 6), (1, 25), (2, 60), (3, 1)]
    with tempfile.TemporaryDirectory() as tmp:
        for ci, (seed, n_img) in enumerate(cases):
            gt, sol = synth(seed, n_img)
            gp, sp = os.path.join(tmp, 'gt.csv'), os.path.join(tmp, 'sol.csv')
            open(gp, 'w').write(gt); open(sp, 'w').write(sol)
            out['case%d_gt' % ci] = np.frombuffer(gt.encode(), np.uint8)
            out['case%d_sol' % ci] = np.frombuffer(sol.encode(), np.uint8)
            for th in ths:
                del used[:]
                ps, rs, m = ref.cal_mAP_fd(gp, sp, th)
                assert used == [6, 7], used                 # the shim served the two slice assignments and nothing else
                out['case%d_th%g_ps' % (ci, th)] = np.asarray(ps, np.float64)
                out['case%d_th%g_rs' % (ci, th)] = np.asarray(rs, np.float64)
                out['case%d_th%g_map' % (ci, th)] = np.float64(m)
                print('case %d (%d images, %d detections) th %.2f: %d points, mAP %.6f' % (ci, n_img, sol.count('\n'), th, len(ps), m))
    out['thresholds'] = np.asarray(ths)
    out['ncases'] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, 'cal_map_fd.npz'), **out)
    print('wrote', os.path.join(HERE, 'cal_map_fd.npz'))


if __name__ == '__main__':
    main()
