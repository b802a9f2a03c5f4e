"""ORACLE -- test infrastructure only ("port" kind: the Keras/TF arithmetic is third-party and
absent from /root/reference, so this restates its documented semantics on torch-CPU).

Network of FaceDetector: Darknet-53 base as wired by FaceDetector.YOLOV3Base
(reference face_detection.py:384-600 over the layer specs yolov3_detect.py:221-267) + the
13x13x6 head (face_detection.py:348-352), trained with loss='mse' (face_detection.py:381) and
keras.optimizers.Adam (face_detection.py:376-379).

PARITY UNPINNED against Keras itself (not installed, no fixtures in the reference): this file is
cross-checked only by hand-computed cases and finite differences (tests/test_net_oracle.py).
Restated Keras 2.2.4 semantics:
  conv   : ZeroPadding2D(1) (symmetric) + Conv2D 'valid', no bias            yd.py:205-211
  BN     : axis -1, eps 1e-3, momentum 0.99; training = batch mean / biased variance;
           moving_var is updated with var * n/(n-(1+eps)) (Keras layer code), plain EMA   yd.py:212
  leaky  : alpha 0.1                                                         yd.py:213
  add    : skip + x after the block's second activation                      fd.py:445,481,...
  head   : Conv2D(6, 3x3, 'same', linear, bias)                              fd.py:348-352
  mse    : mean over every element of (B,G,G,6)                              fd.py:381
  Adam   : lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+1e-7)       SURVEY 8a-9

Flat parameter layout (shared with the library, see include/fv_hotpath.h fv_layer_desc):
  per base layer: kernel OHWI [cout][kh][kw][cin], gamma[cout], beta[cout]; head: kernel OHWI,
  bias[6].  Flat BN state: per base layer moving_mean[cout], moving_var[cout].
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
BN_MOMENTUM = 0.99
LEAKY = 0.1
HEAD_C = 6


def layer_table():
    """[(darknet_idx, k, stride, cin, cout, role)], role in {'plain','res_a','res_b'}; res_b is
    followed by add(skip, x) where skip is the input of the matching res_a."""
    t = [(0, 3, 1, 3, 32, 'plain')]
    idx = 1
    cin = 32
    for cout, nblocks in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        t.append((idx, 3, 2, cin, cout, 'plain')); idx += 1
        for _ in range(nblocks):
            t.append((idx, 1, 1, cout, cout // 2, 'res_a')); idx += 1
            t.append((idx, 3, 1, cout // 2, cout, 'res_b')); idx += 1
            idx += 1  # darknet shortcut layer takes an index
        cin = cout
    return t


def param_layout():
    """-> (entries, n_params, n_state); entry = dict(name, idx, k, s, cin, cout, role, w_off,
    gamma_off, beta_off, mean_off, var_off) (head: bias_off instead of gamma/beta)."""
    ents = []
    off = 0
    soff = 0
    for (idx, k, s, cin, cout, role) in layer_table():
        e = dict(name='conv_%d' % idx, idx=idx, k=k, s=s, cin=cin, cout=cout, role=role, has_bn=True)
        e['w_off'] = off; off += cout * k * k * cin
        e['gamma_off'] = off; off += cout
        e['beta_off'] = off; off += cout
        e['mean_off'] = soff; soff += cout
        e['var_off'] = soff; soff += cout
        ents.append(e)
    e = dict(name='output', idx=-1, k=3, s=1, cin=1024, cout=HEAD_C, role='head', has_bn=False)
    e['w_off'] = off; off += HEAD_C * 9 * 1024
    e['bias_off'] = off; off += HEAD_C
    ents.append(e)
    return ents, off, soff


def init_params(seed=7, dtype=torch.float32):
    """Synthetic weights of SURVEY 8d config 2: kernels ~ N(0, 2/fan_in), gamma 1, beta 0,
    moving mean 0 / var 1; head glorot_uniform, zero bias (Keras default, fd.py:348-352)."""
    ents, n, ns = param_layout()
    g = torch.Generator().manual_seed(seed)
    p = torch.zeros(n, dtype=torch.float64)
    st = torch.zeros(ns, dtype=torch.float64)
    for e in ents:
        k, cin, cout = e['k'], e['cin'], e['cout']
        cnt = cout * k * k * cin
        if e['has_bn']:
            p[e['w_off']:e['w_off'] + cnt] = torch.randn(cnt, generator=g, dtype=torch.float64) * math.sqrt(2.0 / (k * k * cin))
            p[e['gamma_off']:e['gamma_off'] + cout] = 1.0
            st[e['var_off']:e['var_off'] + cout] = 1.0
        else:
            lim = math.sqrt(6.0 / (k * k * cin + k * k * cout))
            p[e['w_off']:e['w_off'] + cnt] = (torch.rand(cnt, generator=g, dtype=torch.float64) * 2 - 1) * lim
    return p.to(dtype), st.to(dtype)


def _conv(x_nchw, w_ohwi, k, s):
    w = w_ohwi.permute(0, 3, 1, 2)  # OIHW
    if k == 3:
        x_nchw = F.pad(x_nchw, (1, 1, 1, 1))
    return F.conv2d(x_nchw, w, stride=s)


def ema_coefficients(step=0, momentum=None):
    """(c_old, c_new) of moving <- c_old*moving + c_new*batch.  step 0: plain EMA.  step t >= 1: the t-th update of TF 1.x
    `assign_moving_average(..., zero_debias=True)` as Keras 2.2.4's K.moving_average_update calls it: zero-initialised
    biased accumulator b_t = m b_{t-1} + (1-m) x_t, moving_t = b_t / (1 - m^t), written in terms of moving_{t-1}."""
    m = BN_MOMENTUM if momentum is None else momentum
    if step <= 0:
        return m, 1 - m
    den = 1.0 - m ** step
    return m * (1.0 - m ** (step - 1)) / den, (1.0 - m) / den


def forward(params, state, x_nhwc, training, update_state=True, return_intermediates=False, positive=None, ema_step=0):
    """x (B,S,S,3) -> (B,S/32,S/32,6).  training=True uses batch statistics and returns the new
    moving state as second value; training=False uses `state` (Keras predict).

    positive: optional list (one bool NHWC tensor per base layer).  LeakyReLU's derivative jumps at 0, so
    two correct evaluations in different precisions can sit on different sides of the kink for the few
    elements within rounding of it -- and then their GRADIENTS differ by O(1) there.  With `positive`
    given, element e of layer l takes slope 1 where positive[l][e] else 0.1, i.e. the function is
    evaluated on the branch the device took (y -> y*slope is continuous across the kink, so the forward
    value moves by less than the rounding that caused the disagreement)."""
    ents, _, _ = param_layout()
    x = x_nhwc.permute(0, 3, 1, 2)
    new_state = state.clone()
    skip = None
    inter = {}
    for li, e in enumerate(ents):
        k, s, cin, cout = e['k'], e['s'], e['cin'], e['cout']
        w = params[e['w_off']:e['w_off'] + cout * k * k * cin].view(cout, k, k, cin)
        if e['role'] == 'head':
            b = params[e['bias_off']:e['bias_off'] + cout]
            x = _conv(x, w, 3, 1) + b.view(1, -1, 1, 1)
            break
        if e['role'] == 'res_a':
            skip = x
        z = _conv(x, w, k, s)
        gamma = params[e['gamma_off']:e['gamma_off'] + cout]
        beta = params[e['beta_off']:e['beta_off'] + cout]
        if training:
            mean = z.mean(dim=(0, 2, 3))
            var = ((z - mean.view(1, -1, 1, 1)) ** 2).mean(dim=(0, 2, 3))
            if update_state:
                n = z.numel() // cout
                with torch.no_grad():
                    mm = state[e['mean_off']:e['mean_off'] + cout]
                    mv = state[e['var_off']:e['var_off'] + cout]
                    c_old, c_new = ema_coefficients(ema_step)
                    new_state[e['mean_off']:e['mean_off'] + cout] = c_old * mm + c_new * mean
                    new_state[e['var_off']:e['var_off'] + cout] = c_old * mv + c_new * var * (n / (n - (1.0 + BN_EPS)))
        else:
            mean = state[e['mean_off']:e['mean_off'] + cout]
            var = state[e['var_off']:e['var_off'] + cout]
        y = (z - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + BN_EPS) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
        if positive is None:
            x = F.leaky_relu(y, LEAKY)
        else:
            pos = positive[li].permute(0, 3, 1, 2)
            x = y * torch.where(pos, torch.ones((), dtype=y.dtype), torch.full((), LEAKY, dtype=y.dtype))
        if e['role'] == 'res_b':
            x = skip + x
        if return_intermediates:
            inter[e['name']] = (z.permute(0, 2, 3, 1), x.permute(0, 2, 3, 1))
    out = x.permute(0, 2, 3, 1).contiguous()
    if return_intermediates:
        return out, new_state, inter
    return out, new_state


def mse(y_pred, y_true):
    return ((y_pred - y_true) ** 2).mean()


def fd_loss(y_pred, y_true, eps=1e-7):
    """The reference's fd_loss (face_detection.py:59-64; defined, never used) with Keras 2.2.4
    K.binary_crossentropy on probabilities (clip to [eps, 1-eps]); returns the mean over cells."""
    def bce(t, o):
        o = torch.clamp(o, eps, 1.0 - eps)
        return -(t * torch.log(o) + (1.0 - t) * torch.log1p(-o))
    o_loss = bce(y_true[..., 0], y_pred[..., 0])
    l2_loss = torch.mean(torch.abs(y_true[..., 1:5] - y_pred[..., 1:5]), dim=-1)
    c_loss = bce(y_true[..., 5], y_pred[..., 5])
    return ((o_loss + l2_loss + c_loss) / 3.0).mean()


def train_step_grads(params, state, x, y_true, positive=None):
    """One fwd + mse + bwd: -> (loss, grads flat, new_state).  positive: see forward()."""
    p = params.clone().requires_grad_(True)
    y, new_state = forward(p, state, x, training=True, positive=positive)
    loss = mse(y, y_true)
    (g,) = torch.autograd.grad(loss, p)
    return loss.detach(), g, new_state


def keras_adam(p, g, m, v, iteration, lr, beta_1, beta_2, decay=0.0, eps=1e-7):
    """Keras 2.2.4 Adam.get_updates restated; iteration = optimizer.iterations
    BEFORE the update (0 for the first step).  Returns (p, m, v)."""
    if decay > 0:
        lr = lr * (1.0 / (1.0 + decay * iteration))
    t = iteration + 1
    lr_t = lr * (math.sqrt(1.0 - beta_2 ** t) / (1.0 - beta_1 ** t))
    m = beta_1 * m + (1.0 - beta_1) * g
    v = beta_2 * v + (1.0 - beta_2) * g * g
    p = p - lr_t * m / (torch.sqrt(v) + eps)
    return p, m, v


def fwd_flops_per_image(image_size=416):
    """2*MAC of base + head (49.050 GFLOP @416)."""
    ents, _, _ = param_layout()
    div = 1
    total = 0
    for e in ents:
        if e['s'] == 2:
            div *= 2
        hw = (image_size // div) ** 2
        total += 2 * hw * e['k'] ** 2 * e['cin'] * e['cout']
    return total


# ----------------------------------------------------------------------------- full YOLOv3 (secondary)
def yolov3_extra_table(nclass_ch=255):
    """Layers 75..105 of make_yolov3_model (yolov3_detect.py:269-308), in execution order:
    (darknet_idx, k, cin, cout, bn_leaky, src) where src names the input tensor:
    'prev', 'base' (13x13x1024 base output), 'route79'/'route91' (branch inputs of the 1x1 before
    UpSampling2D), 'cat61' (concat[upsampled, skip_61]) and 'cat36'."""
    t = []
    c = 1024
    for i, (idx, k, co) in enumerate([(75, 1, 512), (76, 3, 1024), (77, 1, 512), (78, 3, 1024), (79, 1, 512)]):
        t.append((idx, k, c, co, True, 'base' if i == 0 else 'prev')); c = co
    t.append((80, 3, 512, 1024, True, 'prev'))
    t.append((81, 1, 1024, nclass_ch, False, 'prev'))        # yolo_82
    t.append((84, 1, 512, 256, True, 'route79'))
    c = 768
    for i, (idx, k, co) in enumerate([(87, 1, 256), (88, 3, 512), (89, 1, 256), (90, 3, 512), (91, 1, 256)]):
        t.append((idx, k, c, co, True, 'cat61' if i == 0 else 'prev')); c = co
    t.append((92, 3, 256, 512, True, 'prev'))
    t.append((93, 1, 512, nclass_ch, False, 'prev'))         # yolo_94
    t.append((96, 1, 256, 128, True, 'route91'))
    c = 384
    for i, (idx, k, co) in enumerate([(99, 1, 128), (100, 3, 256), (101, 1, 128), (102, 3, 256), (103, 1, 128), (104, 3, 256)]):
        t.append((idx, k, c, co, True, 'cat36' if i == 0 else 'prev')); c = co
    t.append((105, 1, 256, nclass_ch, False, 'prev'))        # yolo_106
    return t


def yolov3_layout(nclass_ch=255):
    """Flat layout of the full model: the 52 base layers exactly as in param_layout() (without the
    FaceDetector head), then the extra layers: kernel OHWI, then gamma/beta (BN layers) or bias."""
    ents, _, _ = param_layout()
    ents = [dict(e) for e in ents[:-1]]
    off = ents[-1]['beta_off'] + ents[-1]['cout']
    soff = ents[-1]['var_off'] + ents[-1]['cout']
    for (idx, k, cin, cout, bn, src) in yolov3_extra_table(nclass_ch):
        e = dict(name='conv_%d' % idx, idx=idx, k=k, s=1, cin=cin, cout=cout, role='extra', has_bn=bn, src=src)
        e['w_off'] = off; off += cout * k * k * cin
        if bn:
            e['gamma_off'] = off; off += cout
            e['beta_off'] = off; off += cout
            e['mean_off'] = soff; soff += cout
            e['var_off'] = soff; soff += cout
        else:
            e['bias_off'] = off; off += cout
        ents.append(e)
    return ents, off, soff


def yolov3_init(seed=11, nclass_ch=255, dtype=torch.float32):
    ents, n, ns = yolov3_layout(nclass_ch)
    g = torch.Generator().manual_seed(seed)
    p = torch.zeros(n, dtype=torch.float64); st = torch.zeros(ns, dtype=torch.float64)
    for e in ents:
        k, cin, cout = e['k'], e['cin'], e['cout']
        cnt = cout * k * k * cin
        p[e['w_off']:e['w_off'] + cnt] = torch.randn(cnt, generator=g, dtype=torch.float64) * math.sqrt(1.0 / (k * k * cin))
        if e['has_bn']:
            p[e['gamma_off']:e['gamma_off'] + cout] = 0.8 + 0.4 * torch.rand(cout, generator=g, dtype=torch.float64)
            p[e['beta_off']:e['beta_off'] + cout] = 0.1 * torch.randn(cout, generator=g, dtype=torch.float64)
            st[e['mean_off']:e['mean_off'] + cout] = 0.1 * torch.randn(cout, generator=g, dtype=torch.float64)
            st[e['var_off']:e['var_off'] + cout] = 0.5 + torch.rand(cout, generator=g, dtype=torch.float64)
        else:
            p[e['bias_off']:e['bias_off'] + cout] = 0.1 * torch.randn(cout, generator=g, dtype=torch.float64)
    return p.to(dtype), st.to(dtype)


def yolov3_forward(params, state, x_nhwc, nclass_ch=255, training=False, positive=None, new_state=None, capture=None):
    """Forward of make_yolov3_model (yolov3_detect.py:217-311): -> [yolo_82, yolo_94, yolo_106] as NHWC
    tensors (B,S/32,S/32,C), (B,S/16,..), (B,S/8,..).  training=False: the reference's inference graph
    (moving statistics).  training=True (the build's extension): batch statistics in every
    BN layer, Keras-style moving-statistics update written into `new_state` if given; positive: forced
    LeakyReLU branches, one bool NHWC tensor per BN layer in layout order (see forward())."""
    ents, _, _ = yolov3_layout(nclass_ch)
    x = x_nhwc.permute(0, 3, 1, 2)
    skip = None
    t = {}
    outs = []
    nbase = 52
    bn_i = [0]

    def block(e, x):
        k, cin, cout = e['k'], e['cin'], e['cout']
        w = params[e['w_off']:e['w_off'] + cout * k * k * cin].view(cout, k, k, cin)
        z = _conv(x, w, k, e['s'])
        if not e['has_bn']:
            return z + params[e['bias_off']:e['bias_off'] + cout].view(1, -1, 1, 1)
        if training:
            mean = z.mean(dim=(0, 2, 3)); var = ((z - mean.view(1, -1, 1, 1)) ** 2).mean(dim=(0, 2, 3))
            if new_state is not None:
                n = z.numel() // cout
                with torch.no_grad():
                    new_state[e['mean_off']:e['mean_off'] + cout] = BN_MOMENTUM * state[e['mean_off']:e['mean_off'] + cout] + (1 - BN_MOMENTUM) * mean
                    new_state[e['var_off']:e['var_off'] + cout] = BN_MOMENTUM * state[e['var_off']:e['var_off'] + cout] \
                        + (1 - BN_MOMENTUM) * var * (n / (n - (1.0 + BN_EPS)))
        else:
            mean = state[e['mean_off']:e['mean_off'] + cout]; var = state[e['var_off']:e['var_off'] + cout]
        y = (z - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + BN_EPS) * params[e['gamma_off']:e['gamma_off'] + cout].view(1, -1, 1, 1) \
            + params[e['beta_off']:e['beta_off'] + cout].view(1, -1, 1, 1)
        i = bn_i[0]; bn_i[0] += 1
        if capture is not None:      # diagnostics: pre-BN output and the statistics of every BN layer
            capture[e['name']] = (z.detach().permute(0, 2, 3, 1), mean.detach(), var.detach())
        if positive is None:
            return F.leaky_relu(y, LEAKY)
        pos = positive[i].permute(0, 3, 1, 2)
        return y * torch.where(pos, torch.ones((), dtype=y.dtype), torch.full((), LEAKY, dtype=y.dtype))

    for li, e in enumerate(ents[:nbase]):
        if e['role'] == 'res_a':
            skip = x
        x = block(e, x)
        if e['role'] == 'res_b':
            x = skip + x
        if e['idx'] == 35:
            t['skip36'] = x      # output of the add after conv_35 (Darknet layer 36)
        if e['idx'] == 60:
            t['skip61'] = x
    t['base'] = x
    prev = x
    for e in ents[nbase:]:
        src = e['src']
        if src == 'prev':
            inp = prev
        elif src == 'base':
            inp = t['base']
        elif src == 'route79':
            inp = t['out79']
        elif src == 'route91':
            inp = t['out91']
        elif src == 'cat61':
            inp = torch.cat([F.interpolate(prev, scale_factor=2, mode='nearest'), t['skip61']], dim=1)
        else:
            inp = torch.cat([F.interpolate(prev, scale_factor=2, mode='nearest'), t['skip36']], dim=1)
        prev = block(e, inp)
        if e['idx'] in (79, 91):
            t['out%d' % e['idx']] = prev
        if not e['has_bn']:
            outs.append(prev.permute(0, 2, 3, 1).contiguous())
    return outs


def yolo_scale_loss(t, y, nclass):
    """The build's three-scale detection loss for ONE scale (the reference defines none; include/fv_hotpath.h
    fv_yolov3_train_step): t, y (B,g,g,3*(5+nclass)); per (cell, anchor)
    (bce(t4,y4) + mean_{k<4}|t_k-y_k| + mean_c bce(t_{5+c},y_{5+c})) / 3 with bce on logits; mean over cells x anchors."""
    B, g = t.shape[0], t.shape[1]
    t = t.reshape(B, g, g, 3, 5 + nclass); y = y.reshape(B, g, g, 3, 5 + nclass)
    bce = lambda a, b: torch.clamp(a, min=0) - a * b + torch.log1p(torch.exp(-a.abs()))
    obj = bce(t[..., 4], y[..., 4])
    box = (t[..., :4] - y[..., :4]).abs().mean(-1)
    cls = bce(t[..., 5:], y[..., 5:]).mean(-1)
    return ((obj + box + cls) / 3.0).mean()


def yolov3_train_step_grads(params, state, x, targets, nclass_ch=255, positive=None):
    """One fwd (training BN) + three-scale loss + bwd: -> (loss, flat grads, new_state)."""
    p = params.clone().requires_grad_(True)
    new_state = state.clone()
    outs = yolov3_forward(p, state, x, nclass_ch, training=True, positive=positive, new_state=new_state)
    nclass = nclass_ch // 3 - 5
    loss = sum(yolo_scale_loss(o, y, nclass) for o, y in zip(outs, targets))
    (g,) = torch.autograd.grad(loss, p)
    return loss.detach(), g, new_state
