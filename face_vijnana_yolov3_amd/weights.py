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
