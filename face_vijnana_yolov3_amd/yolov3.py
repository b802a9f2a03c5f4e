"""Secondary path: the full three-scale YOLOv3 model of the reference (make_yolov3_model,
yolov3_detect.py:217-311) and its decode / NMS chain (yolov3_detect.py:335-444, driver
yolov3_detect.py:_main_ :545-610), on the device.  FaceDetector does not use this; it exists
because SURVEY 8a-17/18 list it as part of the reference's hot-path files."""
import ctypes

import numpy as np
import torch

from ._lib import BUCKET_FN, Context, LayerDesc, lib, ptr

COCO_ANCHORS = [[116, 90, 156, 198, 373, 326], [30, 61, 62, 45, 59, 119], [10, 13, 16, 30, 33, 23]]  # yd.py:560


def yolov3_layer_table(out_channels=255):
    L = lib()
    out = []
    for i in range(L.fv_yolov3_num_layers()):
        d = LayerDesc()
        assert L.fv_yolov3_layer(i, out_channels, ctypes.byref(d)) == 0
        out.append({f: getattr(d, f) for f, _ in LayerDesc._fields_})
    return out


class Yolov3(object):
    def __init__(self, device=0, out_channels=255, ctx=None):
        self.ctx = ctx or Context(device)
        self.dev = torch.device('cuda', self.ctx.device)
        self.out_channels = int(out_channels)
        self.nclass = self.out_channels // 3 - 5
        self.layers = yolov3_layer_table(self.out_channels)
        self.n_params = int(lib().fv_yolov3_param_count(self.out_channels))
        self.n_state = int(lib().fv_yolov3_state_count(self.out_channels))
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.dev)
        self.state = torch.zeros(self.n_state, dtype=torch.float32, device=self.dev)
        self._ws = {}
        self._tws = {}
        self.grads = self.m = self.v = None
        self.iterations = 0
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.bn_zero_debias = False      # Engine.bn_zero_debias: Keras 2.2.4's zero-debiased moving statistics
        self.bn_updates = 0

    def set_params(self, params, state):
        self.params.copy_(torch.as_tensor(params, dtype=torch.float32).reshape(-1))
        self.state.copy_(torch.as_tensor(state, dtype=torch.float32).reshape(-1))

    def init_synthetic(self, seed=7):
        """Random-init weights (no pretrained file offline): BN layers ~ N(0, 2/fan_in), gamma 1, beta 0, moving mean 0 /
        var 1; the three detection convs glorot-uniform with zero bias (as Engine.init_synthetic)."""
        g = torch.Generator(device='cpu').manual_seed(seed)
        p = torch.zeros(self.n_params, dtype=torch.float32); s = torch.zeros(self.n_state, dtype=torch.float32)
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

    def train_flops_per_image(self, S):
        """forward + weight-gradient of every conv + data-gradient of every conv but the first (2 FLOPs per MAC)."""
        f = [2.0 * d['ksize'] ** 2 * d['cin'] * d['cout'] * (S // d['out_div']) ** 2 for d in self.layers]
        return 3.0 * sum(f) - f[0]

    def load_darknet(self, path_or_bytes):
        """Full Darknet yolov3.weights (conv index order 0..105; yd.py:90-121)."""
        from .weights import header_len
        buf = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, 'rb').read()
        data = np.frombuffer(buf, dtype='<f4', offset=header_len(buf))
        p = np.zeros(self.n_params, np.float32); s = np.zeros(self.n_state, np.float32)
        off = 0
        for d in sorted(self.layers, key=lambda d: d['darknet_index']):
            k, cin, cout = d['ksize'], d['cin'], d['cout']
            n = cout * cin * k * k
            if d['has_bn']:
                for dst, o in ((p, d['beta_off']), (p, d['gamma_off']), (s, d['mean_off']), (s, d['var_off'])):
                    dst[o:o + cout] = data[off:off + cout]; off += cout
            else:
                p[d['beta_off']:d['beta_off'] + cout] = data[off:off + cout]; off += cout
            p[d['w_off']:d['w_off'] + n] = data[off:off + n].reshape(cout, cin, k, k).transpose(0, 2, 3, 1).reshape(-1); off += n
        self.set_params(p, s)
        return off

    def predict_device(self, x):
        x = torch.as_tensor(x).to(device=self.dev, dtype=torch.float32).contiguous()
        B, S = x.shape[0], x.shape[1]
        cap = (1 << 29) // (32 * S * S)              # one buffer descriptor per tensor: see Engine.max_infer_batch
        if B > cap >= 1:
            step = cap // 8 * 8 if cap >= 8 else cap
            parts = [self.predict_device(x[i:i + step]) for i in range(0, B, step)]
            return [torch.cat([p[k] for p in parts]) for k in range(3)]
        key = (B, S)
        if key not in self._ws:
            n = int(lib().fv_yolov3_workspace_bytes(B, S, self.out_channels))
            self._ws = {key: torch.empty(n, dtype=torch.uint8, device=self.dev)}
        ws = self._ws[key]
        ys = [torch.empty((B, S // d, S // d, self.out_channels), dtype=torch.float32, device=self.dev) for d in (32, 16, 8)]
        rc = lib().fv_yolov3_forward(self.ctx.handle, ptr(self.params), ptr(self.state), ptr(x), B, S, self.out_channels, ptr(ws),
                                     ws.numel(), ptr(ys[0]), ptr(ys[1]), ptr(ys[2]))
        self.ctx.check(rc, 'fv_yolov3_forward')
        return ys


    # ------------------------------------------------------------------ training (fv_yolov3_train_step)
    def _train_ws(self, B, S):
        key = (B, S)
        if key not in self._tws:
            n = int(lib().fv_yolov3_train_workspace_bytes(B, S, self.out_channels))
            self._tws = {key: torch.empty(n, dtype=torch.uint8, device=self.dev)}
        return self._tws[key]

    def ensure_optimizer(self):
        if self.grads is None:
            self.grads = torch.zeros_like(self.params); self.m = torch.zeros_like(self.params); self.v = torch.zeros_like(self.params)

    def forward_backward(self, x, targets, on_bucket=None, loss_weight=1.0):
        """x (B,S,S,3); targets: three tensors shaped like the outputs, (B,g,g,3*(5+classes)).  Gradients land in
        self.grads; returns the loss (1-element CUDA tensor).  on_bucket(offset, count): called as gradient ranges complete
        (descending offsets), the protocol of Engine.forward_backward -- parallel.DataParallelTrainer drives either."""
        self.ensure_optimizer()
        x = torch.as_tensor(x).to(device=self.dev, dtype=torch.float32).contiguous()
        B, S = x.shape[0], x.shape[1]
        t = [torch.as_tensor(y).to(device=self.dev, dtype=torch.float32).contiguous() for y in targets]
        for y, dv in zip(t, (32, 16, 8)):
            assert y.numel() == B * (S // dv) ** 2 * self.out_channels, tuple(y.shape)
        ws = self._train_ws(B, S)
        cb_error = []
        if on_bucket is not None:
            def _cb(user, off, cnt):      # a ctypes callback swallows exceptions: keep the first, re-raise after the call
                if cb_error:
                    return
                try:
                    on_bucket(int(off), int(cnt))
                except BaseException as e:   # noqa: B902
                    cb_error.append(e)
            cb = BUCKET_FN(_cb)
        else:
            cb = ctypes.cast(None, BUCKET_FN)
        self._bucket_cb = cb
        self.ctx.set_bn_zero_debias_step(self.bn_updates + 1 if self.bn_zero_debias else 0)
        rc = lib().fv_yolov3_train_step(self.ctx.handle, ptr(self.params), ptr(self.state), ptr(x), ptr(t[0]), ptr(t[1]), ptr(t[2]), B, S,
                                        self.out_channels, ptr(ws), ws.numel(), ptr(self.grads), ptr(self._loss), float(loss_weight), cb, None)
        self.ctx.check(rc, 'fv_yolov3_train_step')
        if cb_error:
            raise cb_error[0]
        self.bn_updates += 1
        return self._loss

    def adam_step(self, lr, beta_1, beta_2, decay=0.0, eps=1e-7):
        rc = lib().fv_adam_step(self.ctx.handle, ptr(self.params), ptr(self.grads), ptr(self.m), ptr(self.v), self.n_params,
                                self.iterations, float(lr), float(beta_1), float(beta_2), float(eps), float(decay))
        self.ctx.check(rc, 'fv_adam_step')
        self.iterations += 1

    def train_on_batch(self, x, targets, lr, beta_1, beta_2, decay=0.0):
        loss = self.forward_backward(x, targets)
        self.adam_step(lr, beta_1, beta_2, decay)
        return loss

    def save(self, path):
        """As Engine.save: `*.h5` = HDF5 in Keras' weight layout, one group per layer as `make_yolov3_model().save_weights` names
        them (conv_i / bnorm_i; the detection convs conv_81 / conv_93 / conv_105 carry a bias), Adam state under /fv."""
        d = dict(iterations=np.int64(self.iterations), out_channels=np.int64(self.out_channels))
        if self.m is not None:
            d['adam_m'] = self.m.cpu().numpy(); d['adam_v'] = self.v.cpu().numpy()
        if str(path).endswith('.h5'):
            from . import weights
            weights.write_keras_h5(path, self.layers, self.params.cpu().numpy(), self.state.cpu().numpy(), nested=None, extras=d)
            return
        d = dict(params=self.params.cpu().numpy(), state=self.state.cpu().numpy(), iterations=d['iterations'], out_channels=d['out_channels'])
        if self.m is not None:
            d['m'] = self.m.cpu().numpy(); d['v'] = self.v.cpu().numpy()
        with open(path, 'wb') as f:
            np.savez(f, **d)

    def load(self, path):
        from . import weights
        from .hdf5_lite import is_hdf5, read_hdf5
        if is_hdf5(path):
            datasets, _ = read_hdf5(path)
            if '/fv/out_channels' in datasets and int(datasets['/fv/out_channels']) != self.out_channels:
                raise ValueError('%s holds a model with %d output channels, this one has %d' % (path, int(datasets['/fv/out_channels']), self.out_channels))
            p, st, found = weights.from_keras_datasets(datasets, self.layers, self.n_params, self.n_state)
            missing = sorted(set(weights.expected_keras_tensors(self.layers)) - set(found))
            if missing:
                raise ValueError('%s lacks %d tensors of this model, e.g. %r' % (path, len(missing), missing[:3]))
            self.set_params(torch.from_numpy(p), torch.from_numpy(st))
            self.iterations = int(datasets['/fv/iterations']) if '/fv/iterations' in datasets else 0
            if '/fv/adam_m' in datasets and '/fv/adam_v' in datasets:
                self.ensure_optimizer()
                self.m.copy_(torch.from_numpy(np.asarray(datasets['/fv/adam_m']))); self.v.copy_(torch.from_numpy(np.asarray(datasets['/fv/adam_v'])))
            return
        with open(path, 'rb') as f:
            d = np.load(f)
            if 'out_channels' in d and int(d['out_channels']) != self.out_channels:
                raise ValueError('%s holds a model with %d output channels, this one has %d' % (path, int(d['out_channels']), self.out_channels))
            self.set_params(torch.from_numpy(d['params']), torch.from_numpy(d['state']))
            self.iterations = int(d['iterations'])
            if 'm' in d:
                self.ensure_optimizer()
                self.m.copy_(torch.from_numpy(d['m'])); self.v.copy_(torch.from_numpy(d['v']))

    def load_base(self, params, state):
        """Copy the 52 base layers (the FaceDetector base: same flat layout at the same offsets) from an Engine-shaped pair."""
        nb = self.layers[51]
        n_p = nb['beta_off'] + nb['cout']; n_s = nb['var_off'] + nb['cout']
        self.params[:n_p].copy_(torch.as_tensor(params, dtype=torch.float32).reshape(-1)[:n_p].to(self.dev))
        self.state[:n_s].copy_(torch.as_tensor(state, dtype=torch.float32).reshape(-1)[:n_s].to(self.dev))

    def leaky_slopes_taken(self, B, S):
        """Per BN layer (fv_yolov3_layer order, detection convs skipped): bool tensor, True where the last train step
        took the positive LeakyReLU branch (see Engine.leaky_slopes_taken)."""
        ws = self._train_ws(B, S)
        out = []
        for l, d in enumerate(self.layers):
            if not d['has_bn']:
                continue
            t = []
            for code in (0, 4, 5):
                off, cnt = ctypes.c_size_t(0), ctypes.c_int64(0)
                assert lib().fv_yolov3_train_workspace_tensor(B, S, self.out_channels, l, code, ctypes.byref(off), ctypes.byref(cnt)) == 0
                t.append(ws[off.value:off.value + 4 * cnt.value].view(torch.float32))
            g = S // d['out_div']
            out.append((t[0].view(B, g, g, d['cout']) * t[1] + t[2]) > 0)
        return out


def decode_nms(ctx, y13, y26, y52, image_hw, net_hw=(416, 416), anchors=COCO_ANCHORS, obj_thresh=0.5, nms_thresh=0.45):
    """One image: three (g,g,3*(5+nclass)) float32 CUDA tensors -> dict(boxes (n,4) int32 image
    pixels, objness (n,), classes (n,nclass) with suppressed entries zeroed), reference list order."""
    g = int(y13.shape[-3])
    nclass = int(y13.shape[-1]) // 3 - 5
    cap = g * g + 2 * (2 * g) * (2 * g) + (4 * g) * (4 * g)   # kept anchors: 1 @g, 2 @2g, 1 @4g (skip list)
    dev = y13.device
    boxes = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    obj = torch.empty((cap,), dtype=torch.float32, device=dev)
    cls = torch.empty((cap, nclass), dtype=torch.float32, device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    anc = (ctypes.c_float * 18)(*[float(v) for row in anchors for v in row])
    rc = lib().fv_yolo_decode_nms(ctx.handle, ptr(y13.contiguous()), ptr(y26.contiguous()), ptr(y52.contiguous()), g, nclass, anc,
                                  float(obj_thresh), float(nms_thresh), int(net_hw[0]), int(net_hw[1]), int(image_hw[0]),
                                  int(image_hw[1]), cap, ptr(boxes), ptr(obj), ptr(cls), ptr(cnt))
    ctx.check(rc, 'fv_yolo_decode_nms')
    n = int(cnt.item())
    return dict(boxes=boxes[:n], objness=obj[:n], classes=cls[:n])


def decode_nms_batch(ctx, y13, y26, y52, image_hw, net_hw=(416, 416), anchors=COCO_ANCHORS, obj_thresh=0.5, nms_thresh=0.45):
    """A batch of images of one size: three (B,g,g,3*(5+nclass)) float32 CUDA tensors -> dict of CUDA tensors boxes (B,cap,4)
    int32, objness (B,cap), classes (B,cap,nclass), count (B,) -- ONE launch pair for the batch, no host sync (the caller
    copies the tensors out and reads count[b] entries of image b)."""
    B, g = int(y13.shape[0]), int(y13.shape[-3])
    nclass = int(y13.shape[-1]) // 3 - 5
    cap = g * g + 2 * (2 * g) * (2 * g) + (4 * g) * (4 * g)
    dev = y13.device
    boxes = torch.empty((B, cap, 4), dtype=torch.int32, device=dev)
    obj = torch.empty((B, cap), dtype=torch.float32, device=dev)
    cls = torch.empty((B, cap, nclass), dtype=torch.float32, device=dev)
    cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    anc = (ctypes.c_float * 18)(*[float(v) for row in anchors for v in row])
    rc = lib().fv_yolo_decode_nms_batch(ctx.handle, ptr(y13.contiguous()), ptr(y26.contiguous()), ptr(y52.contiguous()), B, g, nclass, anc,
                                        float(obj_thresh), float(nms_thresh), int(net_hw[0]), int(net_hw[1]), int(image_hw[0]),
                                        int(image_hw[1]), cap, ptr(boxes), ptr(obj), ptr(cls), ptr(cnt))
    ctx.check(rc, 'fv_yolo_decode_nms_batch')
    return dict(boxes=boxes, objness=obj, classes=cls, count=cnt)
