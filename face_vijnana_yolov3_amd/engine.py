"""Device engine behind FaceDetector: owns the flat parameter / optimiser / BN-state vectors
(torch-ROCm tensors used as plain storage) and drives the C-ABI hot path.

It plays the role of the compiled Keras `Model` in the reference (face_detection.py:341-382):
`predict` (fd.py:899), one `fit_generator` step (fd.py:621-627), `save`/`load` (fd.py:630, 337)."""
import ctypes

import numpy as np
import torch

from ._lib import BUCKET_FN, Context, FvError, LayerDesc, lib, ptr, c_void_p

HEAD_C = 6


def layer_table():
    """List of dicts mirroring fv_layer_desc (works without a GPU)."""
    L = lib()
    out = []
    for i in range(L.fv_num_layers()):
        d = LayerDesc()
        assert L.fv_layer(i, ctypes.byref(d)) == 0
        out.append({f: getattr(d, f) for f, _ in LayerDesc._fields_})
    return out


def fwd_flops_per_image(image_size=416):
    """2*MAC of the 52 base convs + the head at one image (49.050 GFLOP at 416)."""
    return sum(2 * (image_size // d['out_div']) ** 2 * d['ksize'] ** 2 * d['cin'] * d['cout'] for d in layer_table())


def train_flops_per_image(image_size=416):
    """forward + every weight-gradient + every data-gradient except conv_0's (146.85 GFLOP at 416)."""
    return 3 * fwd_flops_per_image(image_size) - 2 * image_size * image_size * 27 * 32


class Engine(object):
    def __init__(self, device=0, stream=None):
        self.ctx = Context(device, stream)
        self.dev = torch.device('cuda', device)
        self.layers = layer_table()
        self.n_params = int(lib().fv_param_count())
        self.n_state = int(lib().fv_state_count())
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.dev)
        self.state = torch.zeros(self.n_state, dtype=torch.float32, device=self.dev)
        self.grads = None
        self.m = None
        self.v = None
        self.iterations = 0
        # Keras 2.2.4 / TF 1.x update of the BN moving statistics (zero-debiased, fv_set_bn_zero_debias_step) instead of the plain
        # EMA; bn_updates counts the training steps of THIS object (TF keeps the step in a graph variable that load_model rebuilds)
        self.bn_zero_debias = False
        self.bn_updates = 0
        self._ws = {}
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self._bucket_cb = None

    # ------------------------------------------------------------------ parameters
    def set_params(self, params, state):
        assert params.numel() == self.n_params and state.numel() == self.n_state
        self.params.copy_(torch.as_tensor(params, dtype=torch.float32).reshape(-1))
        self.state.copy_(torch.as_tensor(state, dtype=torch.float32).reshape(-1))

    def init_synthetic(self, seed=7):
        """Random-init weights of the SURVEY 8d config-2 shape (no pretrained file offline):
        kernels ~ N(0, 2/fan_in), gamma 1, beta 0, moving mean 0 / var 1, head glorot-uniform."""
        g = torch.Generator(device='cpu').manual_seed(seed)
        p = torch.zeros(self.n_params, dtype=torch.float32)
        s = torch.zeros(self.n_state, dtype=torch.float32)
        for d in self.layers:
            k, cin, cout = d['ksize'], d['cin'], d['cout']
            n = cout * k * k * cin
            if d['has_bn']:
                p[d['w_off']:d['w_off'] + n] = torch.randn(n, generator=g) * float(np.sqrt(2.0 / (k * k * cin)))
                p[d['gamma_off']:d['gamma_off'] + cout] = 1.0
                s[d['var_off']:d['var_off'] + cout] = 1.0
            else:
                lim = float(np.sqrt(6.0 / (k * k * cin + k * k * cout)))
                p[d['w_off']:d['w_off'] + n] = (torch.rand(n, generator=g) * 2 - 1) * lim
        self.set_params(p, s)

    def _workspace(self, batch, image_size, training):
        key = (batch, image_size, bool(training))
        if key not in self._ws:
            n = int(lib().fv_workspace_bytes(batch, image_size, 1 if training else 0))
            if n == 0:
                raise FvError('unsupported batch/image_size %r' % (key,))
            self._ws = {k: v for k, v in self._ws.items() if k[2] != bool(training)}  # one per mode
            self._ws[key] = torch.empty(n, dtype=torch.uint8, device=self.dev)
        return self._ws[key]

    def _as_input(self, x):
        x = torch.as_tensor(x)
        if x.dtype != torch.float32 or x.device != self.dev:
            x = x.to(device=self.dev, dtype=torch.float32)
        return x.contiguous()

    @staticmethod
    def max_infer_batch(image_size):
        """Largest batch one fv_forward_infer call takes at this image size: the kernels address a tensor through one buffer descriptor
        (2 GiB, 2^29 floats) and the largest activation is the first layer's batch x S x S x 32 output."""
        return (1 << 29) // (32 * int(image_size) * int(image_size))

    # ------------------------------------------------------------------ predict (fd.py:899)
    def predict_device(self, x):
        """x: (B,S,S,3) float in [0,1] -> (B,S/32,S/32,6) float32 CUDA tensor (stream-ordered)."""
        x = self._as_input(x)
        B, S = x.shape[0], x.shape[1]
        assert x.dim() == 4 and x.shape[2] == S and x.shape[3] == 3
        cap = self.max_infer_batch(S)
        if B > cap >= 1:
            # inference is per image (moving statistics): a batch beyond what fv_forward_infer addresses runs in parts
            step = cap // 8 * 8 if cap >= 8 else cap
            return torch.cat([self.predict_device(x[i:i + step]) for i in range(0, B, step)])
        ws = self._workspace(B, S, False)
        y = torch.empty((B, S // 32, S // 32, HEAD_C), dtype=torch.float32, device=self.dev)
        rc = lib().fv_forward_infer(self.ctx.handle, ptr(self.params), ptr(self.state), ptr(x), B, S, ptr(ws), ws.numel(), ptr(y))
        self.ctx.check(rc, 'fv_forward_infer')
        return y

    def predict(self, x):
        return self.predict_device(x).cpu().numpy()

    def predict_base_device(self, x, with_head=False):
        """The base model alone (FaceDetector.YOLOV3Base, fd.py:384-600): x (B,S,S,3) -> the add_23 output (B,S/32,S/32,1024),
        float32 CUDA tensor; with_head=True also returns the head output of the same pass."""
        x = self._as_input(x)
        B, S = x.shape[0], x.shape[1]
        assert x.dim() == 4 and x.shape[2] == S and x.shape[3] == 3
        cap = self.max_infer_batch(S)
        if B > cap >= 1:
            step = cap // 8 * 8 if cap >= 8 else cap
            parts = [self.predict_base_device(x[i:i + step], with_head) for i in range(0, B, step)]
            return (torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])) if with_head else torch.cat(parts)
        ws = self._workspace(B, S, False)
        feat = torch.empty((B, S // 32, S // 32, self.layers[-1]['cin']), dtype=torch.float32, device=self.dev)
        y = torch.empty((B, S // 32, S // 32, HEAD_C), dtype=torch.float32, device=self.dev) if with_head else None
        rc = lib().fv_forward_base(self.ctx.handle, ptr(self.params), ptr(self.state), ptr(x), B, S, ptr(ws), ws.numel(), ptr(feat),
                                   ptr(y) if with_head else c_void_p(None))
        self.ctx.check(rc, 'fv_forward_base')
        return (feat, y) if with_head else feat

    # ------------------------------------------------------------------ training
    def ensure_optimizer(self):
        if self.grads is None:
            self.grads = torch.zeros_like(self.params)
            self.m = torch.zeros_like(self.params)
            self.v = torch.zeros_like(self.params)

    def forward_backward(self, x, y_true, on_bucket=None, loss_weight=1.0):
        """fwd + mse + bwd; gradients land in self.grads; returns the loss as a 1-element CUDA
        tensor (no host sync).  on_bucket(offset, count) is called as gradient ranges complete.
        loss_weight: this slice's share n_r / N of a merged data-parallel batch (fv_train_step): scales the gradients, not the loss."""
        self.ensure_optimizer()
        x = self._as_input(x)
        y_true = self._as_input(y_true)
        B, S = x.shape[0], x.shape[1]
        assert y_true.shape == (B, S // 32, S // 32, HEAD_C), y_true.shape
        ws = self._workspace(B, S, True)
        cb_error = []
        if on_bucket is not None:
            def _cb(user, off, cnt):
                # an exception raised inside a ctypes callback is printed and swallowed: keep the first one
                # and re-raise it once fv_train_step has returned (later ranges are not forwarded)
                if cb_error:
                    return
                try:
                    on_bucket(int(off), int(cnt))
                except BaseException as e:   # noqa: B902
                    cb_error.append(e)
            cb = BUCKET_FN(_cb)
        else:
            cb = ctypes.cast(None, BUCKET_FN)
        self._bucket_cb = cb  # keep alive during the call
        self.ctx.set_bn_zero_debias_step(self.bn_updates + 1 if self.bn_zero_debias else 0)
        rc = lib().fv_train_step(self.ctx.handle, ptr(self.params), ptr(self.state), ptr(x), ptr(y_true), B, S, ptr(ws),
                                 ws.numel(), ptr(self.grads), ptr(self._loss), float(loss_weight), cb, None)
        self.ctx.check(rc, 'fv_train_step')
        if cb_error:
            raise cb_error[0]
        self.bn_updates += 1
        return self._loss

    def train_tensor(self, batch, image_size, layer, which):
        """View into the training workspace after forward_backward (fv_train_workspace_tensor):
        which = 'z' | 'a' | 'mean' | 'invstd' | 'scale' | 'shift' of base layer `layer`."""
        code = {'z': 0, 'a': 1, 'mean': 2, 'invstd': 3, 'scale': 4, 'shift': 5}[which]
        off, cnt = ctypes.c_size_t(0), ctypes.c_int64(0)
        rc = lib().fv_train_workspace_tensor(batch, image_size, layer, code, ctypes.byref(off), ctypes.byref(cnt))
        if rc != 0:
            raise FvError('fv_train_workspace_tensor(%d, %d, %d, %s) failed' % (batch, image_size, layer, which))
        ws = self._workspace(batch, image_size, True)
        t = ws[off.value:off.value + 4 * cnt.value].view(torch.float32)
        d = self.layers[layer]
        if code <= 1:
            g = image_size // d['out_div']
            t = t.view(batch, g, g, d['cout'])
        return t

    def leaky_slopes_taken(self, batch, image_size):
        """Per base layer, a bool tensor [B][H][W][C]: True where the last train step took the
        positive LeakyReLU branch, i.e. fl(fl(z*scale)+shift) > 0 -- the decision every kernel of the
        step makes (the library is built with -ffp-contract=off)."""
        out = []
        for l in range(len(self.layers) - 1):
            z = self.train_tensor(batch, image_size, l, 'z')
            out.append((z * self.train_tensor(batch, image_size, l, 'scale') + self.train_tensor(batch, image_size, l, 'shift')) > 0)
        return out

    def adam_step(self, lr, beta_1, beta_2, decay=0.0, eps=1e-7):
        rc = lib().fv_adam_step(self.ctx.handle, ptr(self.params), ptr(self.grads), ptr(self.m), ptr(self.v), self.n_params,
                                self.iterations, float(lr), float(beta_1), float(beta_2), float(eps), float(decay))
        self.ctx.check(rc, 'fv_adam_step')
        self.iterations += 1

    def train_on_batch(self, x, y_true, lr, beta_1, beta_2, decay=0.0):
        loss = self.forward_backward(x, y_true)
        self.adam_step(lr, beta_1, beta_2, decay)
        return loss

    # ------------------------------------------------------------------ checkpoint (fd.py:630 model.save / fd.py:337 load_model)
    def save(self, path, nested='model_1'):
        """Weights + BN moving statistics + Adam state.  `*.h5` (the reference's MODEL_PATH / yolov3_base.h5 names) is written as a
        real HDF5 file in Keras' weight layout (weights.write_keras_h5 through the pure-Python hdf5_lite: `model.load_weights`
        of the reference's stack and h5py read it), with this build's Adam vectors and step count under /fv; any other
        extension gets the plain .npz of rounds 1-2.  nested=None writes the one-group-per-layer layout of yolov3_base.h5."""
        d = dict(iterations=np.int64(self.iterations))
        if self.m is not None:
            d['adam_m'] = self.m.cpu().numpy(); d['adam_v'] = self.v.cpu().numpy()
        if str(path).endswith('.h5'):
            from . import weights
            weights.write_keras_h5(path, self.layers, self.params.cpu().numpy(), self.state.cpu().numpy(), nested=nested, extras=d)
            return
        d = dict(params=self.params.cpu().numpy(), state=self.state.cpu().numpy(), iterations=d['iterations'])
        if self.m is not None:
            d['m'] = self.m.cpu().numpy(); d['v'] = self.v.cpu().numpy()
        with open(path, 'wb') as f:
            np.savez(f, **d)

    def load(self, path, require_all=True):
        """HDF5 (a Keras weight / model file: the reference's face_detector.h5 or yolov3_base.h5, or one written by save) or the
        .npz of rounds 1-2 -- told apart by the file signature.  require_all=False accepts a file that holds only some layers (the
        base file loaded into the detector: the head keeps its current values)."""
        from . import weights
        from .hdf5_lite import is_hdf5
        if is_hdf5(path):
            p, st, found = None, None, None
            from .hdf5_lite import read_hdf5
            datasets, _ = read_hdf5(path)
            p, st, found = weights.from_keras_datasets(datasets, self.layers, self.n_params, self.n_state,
                                                       self.params.cpu().numpy(), self.state.cpu().numpy())
            missing = sorted(set(weights.expected_keras_tensors(self.layers)) - set(found))
            if missing and require_all:
                raise FvError('%s lacks %d tensors of this model, e.g. %r' % (path, len(missing), missing[:3]))
            self.set_params(torch.from_numpy(p), torch.from_numpy(st))
            self.iterations = int(datasets['/fv/iterations']) if '/fv/iterations' in datasets else 0
            if '/fv/adam_m' in datasets and '/fv/adam_v' in datasets:
                self.ensure_optimizer()
                self.m.copy_(torch.from_numpy(np.asarray(datasets['/fv/adam_m']))); self.v.copy_(torch.from_numpy(np.asarray(datasets['/fv/adam_v'])))
            return
        with open(path, 'rb') as f:
            d = np.load(f)
            self.set_params(torch.from_numpy(d['params']), torch.from_numpy(d['state']))
            self.iterations = int(d['iterations'])
            if 'm' in d:
                self.ensure_optimizer()
                self.m.copy_(torch.from_numpy(d['m'])); self.v.copy_(torch.from_numpy(d['v']))
