"""Mint tests/golden/keras_layout_bigattr.h5 (+ .npz) with libhdf5 (h5py 3.3.0 under /opt/conda/bin/python3.9):

    /opt/conda/bin/python3.9 tests/golden/make_h5_bigattr_fixture.py

A real face_detector.h5 carries its whole architecture as the root attribute `model_config`: for the 250-layer nested Darknet
base (fd.py:341-352) that JSON is ~300 KB.  An attribute of more than 64 KiB does not fit an object-header message, so libhdf5
moves the object's attributes into DENSE storage (a fractal heap indexed by a v2 B-tree, reached through an Attribute Info
message) -- a container shape the small fixtures of make_h5_fixture.py never produce.  This file has it; the datasets sit where
Keras puts them.  The reader must return every dataset bit for bit whatever it does with the attribute."""
import json
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(77)
    layers = []
    for i in range(250):          # the shape of Keras' layer records, synthetic values
        layers.append({'name': 'conv_%d' % i, 'class_name': 'Conv2D', 'config': {
            'name': 'conv_%d' % i, 'trainable': True, 'filters': int(rng.integers(32, 1024)), 'kernel_size': [3, 3], 'strides': [1, 1],
            'padding': 'valid', 'data_format': 'channels_last', 'dilation_rate': [1, 1], 'activation': 'linear', 'use_bias': False,
            'kernel_initializer': {'class_name': 'VarianceScaling', 'config': {'scale': 1.0, 'mode': 'fan_avg', 'distribution': 'uniform', 'seed': None}},
            'bias_initializer': {'class_name': 'Zeros', 'config': {}}, 'kernel_regularizer': None, 'bias_regularizer': None,
            'activity_regularizer': None, 'kernel_constraint': None, 'bias_constraint': None, 'pad': 'x' * int(rng.integers(300, 700))},
            'inbound_nodes': [[['leaky_%d' % (i - 1), 0, 0, {}]]]})
    cfg = json.dumps({'class_name': 'Model', 'config': {'name': 'model_2', 'layers': layers, 'input_layers': [['input1', 0, 0]],
                                                        'output_layers': [['output', 0, 0]]}}).encode('utf8')
    assert len(cfg) > 200 * 1024
    rec = {}
    path = os.path.join(HERE, 'keras_layout_bigattr.h5')
    with h5py.File(path, 'w') as f:
        f.attrs['keras_version'] = '2.2.4'.encode('utf8')
        f.attrs['backend'] = 'tensorflow'.encode('utf8')
        f.attrs['model_config'] = np.void(cfg) if False else cfg          # a scalar fixed-length string attribute, as Keras writes it
        f.attrs['training_config'] = json.dumps({'loss': 'mse'}).encode('utf8')
        mw = f.create_group('model_weights')
        mw.attrs['layer_names'] = np.asarray([b'input1', b'model_1', b'output'])
        mw.attrs['backend'] = b'tensorflow'; mw.attrs['keras_version'] = b'2.2.4'
        mw.create_group('input1').attrs['weight_names'] = np.asarray([], dtype='S1')
        g = mw.create_group('model_1')
        names = []
        for idx, k, cin, cout in [(0, 3, 3, 4), (1, 3, 4, 8)]:
            for wname, shape in [('conv_%d/kernel:0' % idx, (k, k, cin, cout))] + [('bnorm_%d/%s:0' % (idx, n), (cout,)) for n in ('gamma', 'beta', 'moving_mean', 'moving_variance')]:
                v = rng.standard_normal(shape).astype(np.float32)
                g.create_dataset(wname, data=v); names.append(wname.encode())
                rec['/model_weights/model_1/' + wname] = v
        g.attrs['weight_names'] = np.asarray(names)
        o = mw.create_group('output')
        for wname, shape in [('output/kernel:0', (3, 3, 8, 6)), ('output/bias:0', (6,))]:
            v = rng.standard_normal(shape).astype(np.float32)
            o.create_dataset(wname, data=v)
            rec['/model_weights/output/' + wname] = v
        o.attrs['weight_names'] = np.asarray([b'output/kernel:0', b'output/bias:0'])
    rec['model_config_len'] = np.int64(len(cfg))
    rec['model_config_sha1'] = np.frombuffer(__import__('hashlib').sha1(cfg).digest(), np.uint8)
    np.savez(os.path.join(HERE, 'keras_layout_bigattr.npz'), **{k.replace('/', '|'): v for k, v in rec.items()})
    print('wrote', path, os.path.getsize(path), 'bytes; model_config', len(cfg), 'bytes')


if __name__ == '__main__':
    main()
