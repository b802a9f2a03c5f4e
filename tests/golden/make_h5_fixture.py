"""Mint tests/golden/keras_layout_*.h5 + keras_layout.npz with libhdf5 itself (h5py 3.3.0 / HDF5 1.10.6 under
/opt/conda/bin/python3.9 of the build container; the main interpreter has no h5py):

    /opt/conda/bin/python3.9 tests/golden/make_h5_fixture.py

The files follow the layout Keras 2.2.4 writes (keras/engine/saving.py: `_serialize_model` / `save_weights_to_hdf5_group`,
restated here -- Keras itself is not installed): root attributes keras_version / backend / model_config (bytes), group
`model_weights` with the attributes layer_names / backend / keras_version, one group per layer with the attribute weight_names
(arrays of fixed-length byte strings) and one float32 dataset per weight, created under its TF name (`conv_0/kernel:0`, so a nested
group per layer name); a nested `Model` layer (the reference wraps its Darknet base that way, face_detection.py:344-352) holds
all its weights in ONE layer group.  Two files: the FaceDetector layout (`model_1` + `output`) and the base-only layout
(yolov3_base.h5, face_detection.py:596-598).  The arrays are synthetic; what is pinned is the CONTAINER: a pure-Python reader
(face_vijnana_yolov3_amd/hdf5_lite.py) must return exactly these arrays and attributes from files libhdf5 wrote."""
import json
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SPEC = [(0, 3, 3, 4, True), (1, 3, 4, 8, True), (2, 1, 8, 4, True), (3, 3, 4, 8, True)]     # (idx, k, cin, cout, bn)


def save_attr(g, name, values):
    g.attrs[name] = np.asarray([v.encode('utf8') for v in values])


def weights_group(f, layers, rng, rec, prefix):
    save_attr(f, 'layer_names', [n for n, _ in layers])
    f.attrs['backend'] = 'tensorflow'.encode('utf8')
    f.attrs['keras_version'] = '2.2.4'.encode('utf8')
    for lname, wts in layers:
        g = f.create_group(lname)
        save_attr(g, 'weight_names', [n for n, _ in wts])
        for wname, shape in wts:
            val = rng.standard_normal(shape).astype(np.float32) if shape else np.float32(rng.standard_normal())
            d = g.create_dataset(wname, np.shape(val), dtype=np.float32)
            if np.shape(val):
                d[:] = val
            else:
                d[()] = val
            rec[prefix + '/' + lname + '/' + wname] = np.asarray(val)


def base_weights():
    out = []
    for idx, k, cin, cout, bn in SPEC:
        out.append(('conv_%d/kernel:0' % idx, (k, k, cin, cout)))
        if bn:
            out += [('bnorm_%d/%s:0' % (idx, n), (cout,)) for n in ('gamma', 'beta', 'moving_mean', 'moving_variance')]
    return out


def main():
    rng = np.random.default_rng(2024)
    rec = {}
    # 1. face_detector.h5 layout: input1, the nested base model, the head
    with h5py.File(os.path.join(HERE, 'keras_layout_detector.h5'), 'w') as f:
        f.attrs['keras_version'] = '2.2.4'.encode('utf8')
        f.attrs['backend'] = 'tensorflow'.encode('utf8')
        f.attrs['model_config'] = json.dumps({'class_name': 'Model', 'config': {'name': 'model_2', 'layers': ['...']}}).encode('utf8')
        f.attrs['training_config'] = json.dumps({'loss': 'mse', 'optimizer_config': {'class_name': 'Adam'}}).encode('utf8')
        mw = f.create_group('model_weights')
        weights_group(mw, [('input1', []), ('model_1', base_weights()), ('output', [('output/kernel:0', (3, 3, 8, 6)), ('output/bias:0', (6,))])],
                      rng, rec, 'detector/model_weights')
        ow = f.create_group('optimizer_weights')
        save_attr(ow, 'weight_names', ['Adam/iterations:0', 'training/Adam/Variable:0'])
        it = ow.create_dataset('Adam/iterations:0', (), dtype=np.int64); it[()] = 1234
        v = rng.standard_normal((3, 3, 3, 4)).astype(np.float32)
        ow.create_dataset('training/Adam/Variable:0', v.shape, dtype=np.float32)[:] = v
        rec['detector/optimizer_weights/Adam/iterations:0'] = np.int64(1234)
        rec['detector/optimizer_weights/training/Adam/Variable:0'] = v
    # 2. yolov3_base.h5 layout: every Keras layer of the base is its own group (most of them without weights)
    layers = [('input1', [])]
    for idx, k, cin, cout, bn in SPEC:
        if k > 1:
            layers.append(('zero_padding2d_%d' % idx, []))
        layers.append(('conv_%d' % idx, [('conv_%d/kernel:0' % idx, (k, k, cin, cout))]))
        layers.append(('bnorm_%d' % idx, [('bnorm_%d/%s:0' % (idx, n), (cout,)) for n in ('gamma', 'beta', 'moving_mean', 'moving_variance')]))
        layers.append(('leaky_%d' % idx, []))
    layers.append(('add_1', []))
    with h5py.File(os.path.join(HERE, 'keras_layout_base.h5'), 'w') as f:
        f.attrs['keras_version'] = '2.2.4'.encode('utf8')
        f.attrs['backend'] = 'tensorflow'.encode('utf8')
        f.attrs['model_config'] = json.dumps({'class_name': 'Model', 'config': {'name': 'model_1'}}).encode('utf8')
        weights_group(f.create_group('model_weights'), layers, rng, rec, 'base/model_weights')
    # 3. a many-entry group (forces several symbol-table nodes and a two-level B-tree), float64 / int32 / 2-D data, a chunked and
    #    a compact dataset: container features a real file may show
    with h5py.File(os.path.join(HERE, 'keras_layout_wide.h5'), 'w') as f:
        g = f.create_group('model_weights')
        names = ['layer_%03d' % i for i in range(300)]
        save_attr(g, 'layer_names', names)
        for i, n in enumerate(names):
            gg = g.create_group(n)
            if i % 7 == 0:
                a = rng.standard_normal((5, 3)).astype(np.float64 if i % 14 == 0 else np.float32)
                gg.create_dataset('w:0', data=a)
                rec['wide/model_weights/%s/w:0' % n] = a
        c = rng.integers(-100, 100, (40, 6)).astype(np.int32)
        f.create_dataset('chunked', data=c, chunks=(16, 6))
        rec['wide/chunked'] = c
    np.savez(os.path.join(HERE, 'keras_layout.npz'), **{k.replace('/', '|'): v for k, v in rec.items()})
    print('wrote', sorted(os.listdir(HERE)))


if __name__ == '__main__':
    main()
