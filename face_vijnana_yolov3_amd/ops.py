"""Thin Python wrappers over the operator-level C ABI (device tensors in / out).

Used by the parity tests and by anyone composing the kernels differently from engine.py.
All tensors are contiguous float32 CUDA (ROCm) tensors in NHWC / OHWI layout."""
import torch

from ._lib import lib, ptr, c_void_p

NULL = c_void_p(None)


def _p(t):
    return NULL if t is None else ptr(t)


def pack_first_layer(ctx, w_ohwi):
    cout = w_ohwi.shape[0]
    k = w_ohwi[0].numel()
    out = torch.empty((cout, 32), dtype=torch.float32, device=w_ohwi.device)
    ctx.check(lib().fv_pack_first_layer(ctx.handle, ptr(w_ohwi.contiguous()), cout, k, ptr(out)), 'fv_pack_first_layer')
    return out


def conv2d_forward(ctx, x, w, stride=1, scale=None, shift=None, leaky=-1.0, addend=None, stats=False):
    """x (B,H,W,Cin), w OHWI (Cout,k,k,Cin) -> out (B,H/s,W/s,Cout) [, psum, psq]."""
    B, H, W, cin = x.shape
    cout, k = w.shape[0], w.shape[1]
    wd = pack_first_layer(ctx, w) if cin % 32 else w.contiguous()
    out = torch.empty((B, H // stride, W // stride, cout), dtype=torch.float32, device=x.device)
    psum = psq = None
    if stats:
        rows = lib().fv_conv2d_stat_rows(B * (H // stride) * (W // stride))
        psum = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
        psq = torch.empty_like(psum)
    rc = lib().fv_conv2d_forward(ctx.handle, ptr(x.contiguous()), ptr(wd), B, H, W, cin, cout, k, stride, _p(scale),
                                 _p(shift), float(leaky), _p(addend), ptr(out), _p(psum), _p(psq))
    ctx.check(rc, 'fv_conv2d_forward')
    return (out, psum, psq) if stats else out


def transpose_weights(ctx, w, cout_pad=None):
    cout, k, _, cin = w.shape
    cp = cout_pad or cout
    wt = torch.empty((cin, k * k, cp), dtype=torch.float32, device=w.device)
    ctx.check(lib().fv_transpose_weights(ctx.handle, ptr(w.contiguous()), cout, k * k, cin, cp, ptr(wt)), 'fv_transpose_weights')
    return wt


def conv2d_dgrad(ctx, dy, w, in_hw, stride=1, addend=None):
    """dy (B,Ho,Wo,CoutPad), w OHWI -> dx (B,H,W,Cin)."""
    B = dy.shape[0]
    H, W = in_hw
    cout, k, _, cin = w.shape
    wt = transpose_weights(ctx, w, dy.shape[3])
    dx = torch.empty((B, H, W, cin), dtype=torch.float32, device=dy.device)
    rc = lib().fv_conv2d_dgrad(ctx.handle, ptr(dy.contiguous()), ptr(wt), B, H, W, cin, dy.shape[3], k, stride, _p(addend), ptr(dx))
    ctx.check(rc, 'fv_conv2d_dgrad')
    return dx


def conv2d_wgrad(ctx, x, dy, cout, ksize, stride=1):
    """x (B,H,W,Cin), dy (B,Ho,Wo,Ndy>=cout) -> dw OHWI (cout,k,k,Cin)."""
    B, H, W, cin = x.shape
    dw = torch.zeros((cout, ksize, ksize, cin), dtype=torch.float32, device=x.device)
    rc = lib().fv_conv2d_wgrad(ctx.handle, ptr(x.contiguous()), ptr(dy.contiguous()), B, H, W, cin, cout, dy.shape[3], ksize, stride, ptr(dw))
    ctx.check(rc, 'fv_conv2d_wgrad')
    return dw


def bn_finalize(ctx, psum, psq, count, gamma, beta, eps=1e-3, momentum=0.99, moving_mean=None, moving_var=None):
    C = gamma.numel()
    mk = lambda: torch.empty(C, dtype=torch.float32, device=gamma.device)
    mean, invstd, scale, shift = mk(), mk(), mk(), mk()
    rc = lib().fv_bn_finalize(ctx.handle, ptr(psum), ptr(psq), psum.shape[0], C, int(count), ptr(gamma), ptr(beta), eps, momentum,
                              ptr(mean), ptr(invstd), ptr(scale), ptr(shift), _p(moving_mean), _p(moving_var))
    ctx.check(rc, 'fv_bn_finalize')
    return mean, invstd, scale, shift


def bn_act(ctx, z, scale, shift, skip=None, leaky=0.1):
    out = torch.empty_like(z)
    C = z.shape[-1]
    ctx.check(lib().fv_bn_act(ctx.handle, ptr(z), ptr(scale), ptr(shift), _p(skip), ptr(out), z.numel() // C, C, leaky), 'fv_bn_act')
    return out


def bn_bwd(ctx, g, z, scale, shift, mean, invstd, leaky=0.1):
    C = z.shape[-1]
    rows = z.numel() // C
    n = lib().fv_bn_bwd_scratch_floats(rows, C)
    scratch = torch.empty(2 * n, dtype=torch.float32, device=z.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=z.device)
    dgamma = torch.empty_like(dbeta)
    dz = torch.empty_like(z)
    rc = lib().fv_bn_bwd(ctx.handle, ptr(g.contiguous()), ptr(z), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), rows, C, leaky,
                         ptr(scratch), ptr(dbeta), ptr(dgamma), ptr(dz))
    ctx.check(rc, 'fv_bn_bwd')
    return dz, dgamma, dbeta


# ------------------------------------------------------------------ the fused slot forms fv_train_step runs
def stat_slots(C, device):
    """Zeroed [nslot][2][C] float64 accumulators."""
    return torch.zeros((lib().fv_bn_stat_slots(C), 2, C), dtype=torch.float64, device=device)


def conv2d_forward_slots(ctx, x, w, stride, slots):
    B, H, W, cin = x.shape
    cout, k = w.shape[0], w.shape[1]
    wd = pack_first_layer(ctx, w) if cin % 32 else w.contiguous()
    z = torch.empty((B, H // stride, W // stride, cout), dtype=torch.float32, device=x.device)
    rc = lib().fv_conv2d_forward_slots(ctx.handle, ptr(x.contiguous()), ptr(wd), B, H, W, cin, cout, k, stride, ptr(z), ptr(slots),
                                       slots.shape[0])
    ctx.check(rc, 'fv_conv2d_forward_slots')
    return z


def bn_act_slots(ctx, z, slots, gamma, beta, eps=1e-3, momentum=0.99, moving_mean=None, moving_var=None, skip=None, leaky=0.1):
    C = z.shape[-1]
    rows = z.numel() // C
    mk = lambda: torch.empty(C, dtype=torch.float32, device=z.device)
    mean, invstd, scale, shift = mk(), mk(), mk(), mk()
    out = torch.empty_like(z)
    rc = lib().fv_bn_act_slots(ctx.handle, ptr(z), ptr(slots), slots.shape[0], rows, C, ptr(gamma), ptr(beta), eps, momentum, ptr(mean),
                               ptr(invstd), ptr(scale), ptr(shift), _p(moving_mean), _p(moving_var), _p(skip), ptr(out), leaky)
    ctx.check(rc, 'fv_bn_act_slots')
    return out, mean, invstd, scale, shift


def conv2d_dgrad_bnred(ctx, dy, w, in_hw, stride, bn_z, scale, shift, mean, invstd, slots, addend=None, leaky=0.1):
    """dgrad + fused d-beta/d-gamma reduction of the layer that produced the conv input (adds into `slots`)."""
    B = dy.shape[0]
    H, W = in_hw
    cout, k, _, cin = w.shape
    wt = transpose_weights(ctx, w, dy.shape[3])
    dx = torch.empty((B, H, W, cin), dtype=torch.float32, device=dy.device)
    rc = lib().fv_conv2d_dgrad_bnred(ctx.handle, ptr(dy.contiguous()), ptr(wt), B, H, W, cin, dy.shape[3], k, stride, _p(addend), ptr(dx),
                                     ptr(bn_z), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), leaky, ptr(slots), slots.shape[0])
    ctx.check(rc, 'fv_conv2d_dgrad_bnred')
    return dx


def bn_bwd_slots(ctx, g, z, scale, shift, mean, invstd, slots, reduced, leaky=0.1):
    C = z.shape[-1]
    rows = z.numel() // C
    dbeta = torch.empty(C, dtype=torch.float32, device=z.device)
    dgamma = torch.empty_like(dbeta)
    dz = torch.empty_like(z)
    rc = lib().fv_bn_bwd_slots(ctx.handle, ptr(g.contiguous()), ptr(z), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), rows, C, leaky,
                               ptr(slots), slots.shape[0], 1 if reduced else 0, ptr(dbeta), ptr(dgamma), ptr(dz))
    ctx.check(rc, 'fv_bn_bwd_slots')
    return dz, dgamma, dbeta


def mse_loss_grad(ctx, yp, yt, c_pad=32):
    C = yp.shape[-1]
    rows = yp.numel() // C
    loss = torch.empty(1, dtype=torch.float32, device=yp.device)
    dy = torch.empty((rows, c_pad), dtype=torch.float32, device=yp.device)
    db = torch.empty(C, dtype=torch.float32, device=yp.device)
    rc = lib().fv_mse_loss_grad(ctx.handle, ptr(yp.contiguous()), ptr(yt.contiguous()), rows, C, c_pad, ptr(loss), ptr(dy), ptr(db))
    ctx.check(rc, 'fv_mse_loss_grad')
    return loss, dy, db


def fd_loss_grad(ctx, yp, yt, c_pad=32):
    """The reference's unused fd_loss (face_detection.py:59-64) and its gradient."""
    cells = yp.numel() // 6
    loss = torch.empty(1, dtype=torch.float32, device=yp.device)
    dy = torch.empty((cells, c_pad), dtype=torch.float32, device=yp.device)
    ctx.check(lib().fv_fd_loss_grad(ctx.handle, ptr(yp.contiguous()), ptr(yt.contiguous()), cells, c_pad, ptr(loss), ptr(dy)), 'fv_fd_loss_grad')
    return loss, dy


def adam_step(ctx, p, g, m, v, iteration, lr, beta_1, beta_2, eps=1e-7, decay=0.0):
    rc = lib().fv_adam_step(ctx.handle, ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), int(iteration), float(lr), float(beta_1),
                            float(beta_2), float(eps), float(decay))
    ctx.check(rc, 'fv_adam_step')
