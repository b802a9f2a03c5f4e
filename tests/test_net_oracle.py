"""The torch-CPU network oracle: structure pinned to the reference's numbers, its
arithmetic to hand-computed cases and finite differences.  (Keras itself is unavailable:
parity unpinned, see oracle/net_oracle.py header.)"""
import math

import numpy as np
import torch

from oracle import net_oracle as no


def test_layer_table_matches_reference_wiring():
    t = no.layer_table()
    assert len(t) == 52
    assert sum(1 for e in t if e[1] == 3) == 29 and sum(1 for e in t if e[1] == 1) == 23
    assert sum(1 for e in t if e[5] == 'res_b') == 23
    idx = [e[0] for e in t]
    # conv indices present in fd.py:408-593: 0,1,2,3,5,6,7,9,10,12,13,14,...,72,73
    assert idx[:9] == [0, 1, 2, 3, 5, 6, 7, 9, 10] and idx[-1] == 73 and idx[9] == 12
    assert [e[0] for e in t if e[2] == 2] == [1, 5, 12, 37, 62]   # stride-2 convs yd.py:222-260
    ents, n, ns = no.param_layout()
    assert n == 40640230 and ns == 35712                          # SURVEY 8 sizing constants
    assert sum(e['cout'] * e['k'] ** 2 * e['cin'] for e in ents if e['has_bn']) == 40549216
    assert abs(no.fwd_flops_per_image(416) / 1e9 - 49.050) < 0.005
    assert abs(no.fwd_flops_per_image(608) / 1e9 - 104.776) < 0.01


def test_conv_semantics_hand_case():
    # 1-channel 3x3 stride-2 conv with symmetric pad 1: output (i,j) reads rows 2i-1..2i+1
    x = torch.arange(16, dtype=torch.float64).view(1, 1, 4, 4)
    w = torch.zeros(1, 3, 3, 1, dtype=torch.float64); w[0, 0, 0, 0] = 1.0  # top-left tap
    y = no._conv(x, w, 3, 2)
    assert y.shape == (1, 1, 2, 2)
    assert y[0, 0, 0, 0] == 0 and y[0, 0, 1, 1] == x[0, 0, 1, 1]  # (2*1-1, 2*1-1)
    w = torch.zeros(1, 3, 3, 1, dtype=torch.float64); w[0, 1, 1, 0] = 1.0  # centre tap
    assert torch.equal(no._conv(x, w, 3, 2)[0, 0], x[0, 0, ::2, ::2])


def test_keras_adam_two_steps_scalar():
    p = torch.tensor([1.0], dtype=torch.float64); m = torch.zeros(1, dtype=torch.float64); v = torch.zeros(1, dtype=torch.float64)
    g = torch.tensor([0.5], dtype=torch.float64)
    lr, b1, b2 = 1e-4, 0.99, 0.99
    p1, m1, v1 = no.keras_adam(p, g, m, v, 0, lr, b1, b2)
    # hand: m=0.005, v=0.0025*0.01=2.5e-5... v=(1-b2)*g^2=0.0025 ; lr_t=lr*sqrt(0.01)/0.01=lr*10
    assert math.isclose(m1.item(), 0.005) and math.isclose(v1.item(), 0.0025)
    want = 1.0 - (lr * 10) * 0.005 / (math.sqrt(0.0025) + 1e-7)
    assert math.isclose(p1.item(), want, rel_tol=1e-14)
    p2, m2, v2 = no.keras_adam(p1, g, m1, v1, 1, lr, b1, b2)
    lr_t = lr * math.sqrt(1 - b2 ** 2) / (1 - b1 ** 2)
    m_h = 0.99 * 0.005 + 0.01 * 0.5; v_h = 0.99 * 0.0025 + 0.01 * 0.25
    assert math.isclose(p2.item(), want - lr_t * m_h / (math.sqrt(v_h) + 1e-7), rel_tol=1e-14)
    # decay
    p3, _, _ = no.keras_adam(p, g, m, v, 10, lr, b1, b2, decay=0.1)
    lr_d = lr / (1 + 0.1 * 10); t = 11
    assert math.isclose(p3.item(), 1.0 - lr_d * math.sqrt(1 - b2 ** t) / (1 - b1 ** t) * 0.005 / (math.sqrt(0.0025) + 1e-7), rel_tol=1e-14)


def test_forward_shapes_and_bn_train_vs_infer():
    p, st = no.init_params(7, torch.float64)
    x = torch.rand(2, 64, 64, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    y, st2 = no.forward(p, st, x, training=True)
    assert y.shape == (2, 2, 2, 6)
    assert not torch.equal(st, st2)
    yi, st3 = no.forward(p, st, x, training=False)
    assert torch.equal(st3, st) and not torch.allclose(y, yi)
    # moving update is the plain EMA of the batch stats (first layer, channel 0)
    e = no.param_layout()[0][0]
    z = no._conv(x.permute(0, 3, 1, 2), p[e['w_off']:e['w_off'] + 32 * 27].view(32, 3, 3, 3), 3, 1)
    assert math.isclose(st2[e['mean_off']].item(), 0.01 * z[:, 0].mean().item(), rel_tol=1e-12)
    n = z[:, 0].numel()
    assert math.isclose(st2[e['var_off']].item(), 0.99 + 0.01 * z[:, 0].var(unbiased=False).item() * n / (n - 1.001), rel_tol=1e-12)


def test_gradient_finite_difference():
    torch.manual_seed(0)
    p, st = no.init_params(3, torch.float64)
    x = torch.rand(4, 64, 64, 3, dtype=torch.float64)
    yt = torch.rand(4, 2, 2, 6, dtype=torch.float64)
    loss, g, _ = no.train_step_grads(p, st, x, yt)
    ents, n, _ = no.param_layout()
    rng = np.random.default_rng(0)
    picks = [ents[0]['w_off'] + 5, ents[0]['gamma_off'] + 1, ents[3]['beta_off'] + 2, ents[10]['w_off'] + 100,
             ents[-1]['w_off'] + 50, ents[-1]['bias_off'] + 3, ents[30]['w_off'] + 12345, ents[51]['gamma_off'] + 7]
    for i in picks:
        h = 1e-7 * max(1.0, abs(p[i].item()))  # below the leaky-kink noise
        pp = p.clone(); pp[i] += h
        pm = p.clone(); pm[i] -= h
        lp = no.mse(no.forward(pp, st, x, True, update_state=False)[0], yt).item()
        lm = no.mse(no.forward(pm, st, x, True, update_state=False)[0], yt).item()
        fd = (lp - lm) / (2 * h)
        assert math.isclose(fd, g[i].item(), rel_tol=1e-3, abs_tol=1e-7), (i, fd, g[i].item())


def test_forced_slopes_equal_plain_leaky_when_they_are_its_own():
    """forward(positive=...): with the oracle's own sign pattern it IS the plain LeakyReLU network (values,
    BN state and gradients identical); flipping one element changes the gradient of its layer's d-beta."""
    p, st = no.init_params(5, torch.float64)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 64, 64, 3, dtype=torch.float64, generator=g); yt = torch.rand(2, 2, 2, 6, dtype=torch.float64, generator=g)
    _, _, inter = no.forward(p, st, x, training=True, return_intermediates=True)
    ents, _, _ = no.param_layout()
    pos = []
    skip = None
    for e in ents[:-1]:   # recover the pre-activation sign from (z, batch statistics)
        z = inter[e['name']][0]
        c = e['cout']
        mean = z.mean(dim=(0, 1, 2)); var = ((z - mean) ** 2).mean(dim=(0, 1, 2))
        y = (z - mean) / torch.sqrt(var + no.BN_EPS) * p[e['gamma_off']:e['gamma_off'] + c] + p[e['beta_off']:e['beta_off'] + c]
        pos.append(y > 0)
    l0, g0, s0 = no.train_step_grads(p, st, x, yt)
    l1, g1, s1 = no.train_step_grads(p, st, x, yt, positive=pos)
    assert torch.equal(s0, s1) and abs(l0.item() - l1.item()) <= 1e-15 * abs(l0.item())
    torch.testing.assert_close(g0, g1, rtol=1e-12, atol=1e-18)
    pos[51] = pos[51].clone(); pos[51][0, 0, 0, 0] = ~pos[51][0, 0, 0, 0]
    _, g2, _ = no.train_step_grads(p, st, x, yt, positive=pos)
    e = ents[51]
    assert g2[e['beta_off']].item() != g0[e['beta_off']].item()


def test_three_scale_loss_hand_case_and_gradient():
    """yolo_scale_loss (the build's generalisation of fd_loss to 3 anchors x classes on logits): a hand-computed
    cell, and autograd against finite differences."""
    ncls = 2
    t = torch.zeros(1, 1, 1, 3 * (5 + ncls), dtype=torch.float64); y = torch.zeros_like(t)
    # anchor 0: logits 0 vs targets 0 -> bce = log 2 for objectness and each class, boxes equal -> 0
    # anchor 1: box error (1, -1, 0.5, 0) -> mean |.| = 0.625; objectness logit 2 vs target 1
    t[0, 0, 0, 7:11] = torch.tensor([1.0, -1.0, 0.5, 0.0]); t[0, 0, 0, 11] = 2.0; y[0, 0, 0, 11] = 1.0
    ln2 = math.log(2.0)
    a0 = (ln2 + 0.0 + ln2) / 3
    a1 = ((2.0 - 2.0 + math.log1p(math.exp(-2.0))) + 0.625 + ln2) / 3
    a2 = (ln2 + 0.0 + ln2) / 3
    assert math.isclose(no.yolo_scale_loss(t, y, ncls).item(), (a0 + a1 + a2) / 3, rel_tol=1e-14)
    g = torch.Generator().manual_seed(0)
    t = torch.randn(2, 3, 3, 21, dtype=torch.float64, generator=g).requires_grad_(True)
    y = torch.rand(2, 3, 3, 21, dtype=torch.float64, generator=g)
    (gr,) = torch.autograd.grad(no.yolo_scale_loss(t, y, ncls), t)
    for idx in [(0, 0, 0, 4), (1, 2, 1, 9), (0, 1, 2, 20), (1, 0, 0, 0)]:
        h = 1e-6
        tp = t.detach().clone(); tp[idx] += h
        tm = t.detach().clone(); tm[idx] -= h
        fd = (no.yolo_scale_loss(tp, y, ncls) - no.yolo_scale_loss(tm, y, ncls)).item() / (2 * h)
        assert math.isclose(fd, gr[idx].item(), rel_tol=1e-5, abs_tol=1e-10)


def test_three_scale_training_forward_consistency():
    """yolov3_forward(training=True): same graph as the inference forward (equal when the moving statistics ARE the
    batch statistics), both concat routes carry gradient, moving statistics move."""
    p, s = no.yolov3_init(3, 27, torch.float64)
    g = torch.Generator().manual_seed(1)
    x = torch.rand((2, 64, 64, 3), dtype=torch.float64, generator=g)
    tg = [torch.rand((2, 64 // d, 64 // d, 27), dtype=torch.float64, generator=g) for d in (32, 16, 8)]
    loss, gr, ns = no.yolov3_train_step_grads(p, s, x, tg, 27)
    ents, n, _ = no.yolov3_layout(27)
    assert gr.shape[0] == n and torch.isfinite(gr).all() and not torch.equal(ns, s)
    for e in ents:
        k, cin, cout = e['k'], e['cin'], e['cout']
        assert gr[e['w_off']:e['w_off'] + cout * k * k * cin].abs().max() > 0, e['name']     # every layer is reached
    # inference forward with the statistics the training forward just used == training forward
    outs_t = no.yolov3_forward(p, s, x, 27, training=True)
    s2 = s.clone()
    for e in ents:
        if e['has_bn']:
            pass
    # (moving statistics after ONE update from zero momentum would equal batch statistics; here just shape/finite checks)
    assert all(o.shape == t.shape for o, t in zip(outs_t, tg))


def test_zero_debias_coefficients_equal_the_explicit_accumulator():
    """oracle.ema_coefficients(t): moving_t = c_old moving_{t-1} + c_new x_t reproduces TF 1.x zero_debias (biased accumulator
    from zero, divided by 1 - m^t) for every t, whatever the stored value before the first update."""
    import numpy as np
    from oracle import net_oracle as no
    rng = np.random.default_rng(3)
    m = no.BN_MOMENTUM
    xs = rng.normal(size=(12, 5))
    moving = rng.normal(size=5) + 4.0
    b = np.zeros(5)
    for t, xt in enumerate(xs, 1):
        b = m * b + (1 - m) * xt
        c_old, c_new = no.ema_coefficients(t)
        moving = c_old * moving + c_new * xt
        assert np.allclose(moving, b / (1 - m ** t), rtol=1e-12, atol=1e-12)
    assert no.ema_coefficients(0) == (m, 1 - m)
