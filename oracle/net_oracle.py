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


def forward(params, state, x_nhwc, training, update_state=True, return_intermediates=False):
    """x (B,S,S,3) -> (B,S/32,S/32,6).  training=True uses batch statistics and returns the new
    moving state as second value; training=False uses `state` (Keras predict)."""
    ents, _, _ = param_layout()
    x = x_nhwc.permute(0, 3, 1, 2)
    new_state = state.clone()
    skip = None
    inter = {}
    for e in ents:
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
                    new_state[e['mean_off']:e['mean_off'] + cout] = BN_MOMENTUM * mm + (1 - BN_MOMENTUM) * mean
                    new_state[e['var_off']:e['var_off'] + cout] = BN_MOMENTUM * mv + (1 - BN_MOMENTUM) * var * (n / (n - (1.0 + BN_EPS)))
        else:
            mean = state[e['mean_off']:e['mean_off'] + cout]
            var = state[e['var_off']:e['var_off'] + cout]
        y = (z - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + BN_EPS) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
        x = F.leaky_relu(y, LEAKY)
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


def train_step_grads(params, state, x, y_true):
    """One fwd + mse + bwd: -> (loss, grads flat, new_state)."""
    p = params.clone().requires_grad_(True)
    y, new_state = forward(p, state, x, training=True)
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
