"""Darknet `.weights` <-> flat parameter vector.

Format restated from the reference's WeightReader (yolov3_detect.py:67-124): header = int32
major, minor, revision, then 8 more bytes if major*10+minor >= 2 (both < 1000) else 4; body =
float32 stream; per conv in index order: [beta, gamma, mean, var] (BN layers) or [bias], then the
kernel stored (cout, cin, kh, kw).  FaceDetector only consumes convs 0..73 (the Darknet-53 base).
Our flat layout stores kernels OHWI; BN moving stats go to the state vector."""
import struct

import numpy as np


def header_len(buf):
    major, minor, _ = struct.unpack_from('iii', buf, 0)
    return 12 + (8 if (major * 10 + minor) >= 2 and major < 1000 and minor < 1000 else 4)


def read_darknet_base(path_or_bytes, layers, n_params, n_state):
    """-> (params float32[n_params], state float32[n_state]) with the base layers filled from the
    file; the head stays zero (Keras initialises it separately, face_detection.py:348-352)."""
    buf = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, 'rb').read()
    data = np.frombuffer(buf, dtype='<f4', offset=header_len(buf))
    params = np.zeros(n_params, np.float32)
    state = np.zeros(n_state, np.float32)
    off = 0
    for d in layers:
        if not d['has_bn']:
            continue
        k, cin, cout = d['ksize'], d['cin'], d['cout']
        n = cout * cin * k * k
        if off + 4 * cout + n > data.size:
            raise ValueError('darknet weights file too short at conv_%d' % d['darknet_index'])
        beta = data[off:off + cout]; off += cout
        gamma = data[off:off + cout]; off += cout
        mean = data[off:off + cout]; off += cout
        var = data[off:off + cout]; off += cout
        kern = data[off:off + n].reshape(cout, cin, k, k); off += n
        params[d['w_off']:d['w_off'] + n] = kern.transpose(0, 2, 3, 1).reshape(-1)   # OIHW -> OHWI
        params[d['gamma_off']:d['gamma_off'] + cout] = gamma
        params[d['beta_off']:d['beta_off'] + cout] = beta
        state[d['mean_off']:d['mean_off'] + cout] = mean
        state[d['var_off']:d['var_off'] + cout] = var
    return params, state


def write_darknet_base(path, layers, params, state, major=0, minor=2, revision=0, seen=0):
    """Inverse of read_darknet_base (base layers only) -- lets synthetic weights flow through
    the same file format the reference consumes."""
    params = np.asarray(params, np.float32); state = np.asarray(state, np.float32)
    with open(path, 'wb') as f:
        f.write(struct.pack('iii', major, minor, revision))
        f.write(struct.pack('q', seen) if (major * 10 + minor) >= 2 else struct.pack('i', seen))
        for d in layers:
            if not d['has_bn']:
                continue
            k, cin, cout = d['ksize'], d['cin'], d['cout']
            n = cout * cin * k * k
            f.write(params[d['beta_off']:d['beta_off'] + cout].tobytes())
            f.write(params[d['gamma_off']:d['gamma_off'] + cout].tobytes())
            f.write(state[d['mean_off']:d['mean_off'] + cout].tobytes())
            f.write(state[d['var_off']:d['var_off'] + cout].tobytes())
            f.write(params[d['w_off']:d['w_off'] + n].reshape(cout, k, k, cin).transpose(0, 3, 1, 2).tobytes())


def keras_weights(layers, params, state):
    """{'conv_i': [kernel HWIO (, bias)], 'bnorm_i': [gamma, beta, mean, var]} -- the arrays Keras'
    set_weights would receive (yolov3_detect.py:101-119)."""
    out = {}
    params = np.asarray(params); state = np.asarray(state)
    for d in layers:
        k, cin, cout = d['ksize'], d['cin'], d['cout']
        kern = params[d['w_off']:d['w_off'] + cout * k * k * cin].reshape(cout, k, k, cin).transpose(1, 2, 3, 0)
        if d['has_bn']:
            i = d['darknet_index']
            out['conv_%d' % i] = [kern]
            out['bnorm_%d' % i] = [params[d['gamma_off']:d['gamma_off'] + cout], params[d['beta_off']:d['beta_off'] + cout],
                                   state[d['mean_off']:d['mean_off'] + cout], state[d['var_off']:d['var_off'] + cout]]
        else:
            out['output'] = [kern, params[d['beta_off']:d['beta_off'] + cout]]
    return out


# ----------------------------------------------------------------------------- Keras HDF5 weight files
# Layout restated from Keras 2.2.4 keras/engine/saving.py (`save_weights_to_hdf5_group`; Keras is not installed here):
#   /model_weights            attrs layer_names, backend, keras_version
#   /model_weights/<layer>    attr weight_names; one dataset per weight under its TF name, e.g.
#       face_detector.h5 :  /model_weights/model_1/conv_0/kernel:0   (the Darknet base is ONE nested-Model layer, fd.py:344-352)
#                           /model_weights/output/output/kernel:0, .../bias:0
#       yolov3_base.h5   :  /model_weights/conv_0/conv_0/kernel:0, /model_weights/bnorm_0/bnorm_0/gamma:0 ... (fd.py:596-598)
# Conv kernels are HWIO there and OHWI in the flat vector; BatchNormalization weights are gamma, beta, moving_mean, moving_variance.
import re

_KERAS_WEIGHT = re.compile(r'/(?:(conv|bnorm)_(\d+)|(output))(?:_\d+)?/(kernel|bias|gamma|beta|moving_mean|moving_variance)(?:_\d+)?:0$')


def from_keras_datasets(datasets, layers, n_params, n_state, params=None, state=None):
    """{'/model_weights/.../conv_5/kernel:0': array} -> (params, state, found) for the layers of `layers` (fv_layer /
    fv_yolov3_layer dicts).  Tensors of layers the file does not hold keep the values of `params` / `state` (zeros when not
    given); `found` lists what was filled, e.g. ('conv', 5, 'kernel')."""
    params = np.zeros(n_params, np.float32) if params is None else np.array(params, np.float32, copy=True)
    state = np.zeros(n_state, np.float32) if state is None else np.array(state, np.float32, copy=True)
    by_idx = {d['darknet_index']: d for d in layers}
    found = []
    for path in sorted(datasets):
        m = _KERAS_WEIGHT.search(path)
        if not m:
            continue
        kind, idx, out, what = m.group(1), m.group(2), m.group(3), m.group(4)
        d = by_idx.get(-1 if out else int(idx))
        if d is None:
            continue
        a = np.asarray(datasets[path], np.float32)
        k, cin, cout = d['ksize'], d['cin'], d['cout']
        if what == 'kernel':
            if a.shape != (k, k, cin, cout):
                raise ValueError('%s has shape %r, the layer expects %r' % (path, a.shape, (k, k, cin, cout)))
            params[d['w_off']:d['w_off'] + a.size] = a.transpose(3, 0, 1, 2).reshape(-1)          # HWIO -> OHWI
        elif a.shape != (cout,):
            raise ValueError('%s has shape %r, the layer expects (%d,)' % (path, a.shape, cout))
        elif what == 'bias':
            if d['has_bn']:
                continue
            params[d['beta_off']:d['beta_off'] + cout] = a
        elif not d['has_bn']:
            continue
        elif what == 'gamma':
            params[d['gamma_off']:d['gamma_off'] + cout] = a
        elif what == 'beta':
            params[d['beta_off']:d['beta_off'] + cout] = a
        elif what == 'moving_mean':
            state[d['mean_off']:d['mean_off'] + cout] = a
        else:
            state[d['var_off']:d['var_off'] + cout] = a
        found.append(('output' if out else kind, -1 if out else int(idx), what))
    return params, state, found


def expected_keras_tensors(layers):
    out = []
    for d in layers:
        i = d['darknet_index']
        if d['has_bn']:
            out.append(('conv', i, 'kernel'))
            out += [('bnorm', i, w) for w in ('gamma', 'beta', 'moving_mean', 'moving_variance')]
        else:
            nm = 'output' if i < 0 else 'conv'
            out += [(nm, i, 'kernel'), (nm, i, 'bias')]
    return out


def read_keras_h5(path, layers, n_params, n_state, require_all=True):
    """A Keras weight / model file (the reference's face_detector.h5 or yolov3_base.h5, or one written by write_keras_h5) ->
    (params, state, extras); extras holds this build's own datasets under /fv (Adam state), if any."""
    from .hdf5_lite import read_hdf5
    datasets, _attrs = read_hdf5(path)
    params, state, found = from_keras_datasets(datasets, layers, n_params, n_state)
    if require_all:
        missing = sorted(set(expected_keras_tensors(layers)) - set(found))
        if missing:
            raise ValueError('%s lacks %d tensors of this model, e.g. %r' % (path, len(missing), missing[:3]))
    extras = {k[len('/fv/'):]: v for k, v in datasets.items() if k.startswith('/fv/')}
    return params, state, extras


def write_keras_h5(path, layers, params, state, nested='model_1', extras=None):
    """Flat vectors -> an HDF5 file in Keras' weight layout (readable by h5py / `model.load_weights`; no `model_config`, so not by
    `load_model`).  nested: name of the nested-Model layer that holds every BN layer (face_detector.h5), or None for one group per
    Keras layer (yolov3_base.h5).  extras: {name: array} stored under /fv (this build's Adam state)."""
    from .hdf5_lite import write_hdf5
    kw = keras_weights(layers, params, state)
    data, attrs = {}, {}
    groups = {}                                           # layer group -> [(weight name, array)]
    for d in layers:
        i = d['darknet_index']
        if d['has_bn']:
            conv = [('conv_%d/kernel:0' % i, kw['conv_%d' % i][0])]
            bn = [('bnorm_%d/%s:0' % (i, n), kw['bnorm_%d' % i][j]) for j, n in enumerate(('gamma', 'beta', 'moving_mean', 'moving_variance'))]
            if nested:
                groups.setdefault(nested, []).extend(conv + bn)
            else:
                groups['conv_%d' % i] = conv
                groups['bnorm_%d' % i] = bn
        else:
            name = 'output' if i < 0 else 'conv_%d' % i
            k, cin, cout = d['ksize'], d['cin'], d['cout']
            kern = np.asarray(params)[d['w_off']:d['w_off'] + cout * k * k * cin].reshape(cout, k, k, cin).transpose(1, 2, 3, 0)
            bias = np.asarray(params)[d['beta_off']:d['beta_off'] + cout]
            groups[name] = [('%s/kernel:0' % name, kern), ('%s/bias:0' % name, bias)]
    names = (['input1'] if nested else []) + list(groups)
    fixed = lambda xs: np.array([x.encode('utf8') for x in xs]) if xs else np.zeros((0,), 'S1')
    attrs['/'] = {'keras_version': b'2.2.4', 'backend': b'tensorflow'}
    attrs['/model_weights'] = {'layer_names': fixed(names), 'backend': b'tensorflow', 'keras_version': b'2.2.4'}
    if nested:
        attrs['/model_weights/input1'] = {'weight_names': fixed([])}
    for g, wts in groups.items():
        attrs['/model_weights/' + g] = {'weight_names': fixed([n for n, _ in wts])}
        for n, a in wts:
            data['/model_weights/%s/%s' % (g, n)] = np.ascontiguousarray(a, dtype=np.float32)
    for k, v in (extras or {}).items():
        data['/fv/' + k] = np.asarray(v)
    write_hdf5(path, data, attrs)
