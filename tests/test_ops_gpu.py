"""GPU parity of every operator-level kernel (through the C ABI) against a float64 torch-CPU
restatement of the same Keras op (oracle/net_oracle.py conventions).

Tolerance: the kernels compute in exact fp32 (v_mfma_f32_32x32x2_f32 = fmaf chain); against the
float64 reference the error is bounded by ~K * 2^-24 * sum|a||b|, so we assert
|got - ref| <= 2e-6 * (|a| conv |b|) + 1e-6 -- i.e. a few fp32 ulps of the absolute-value
convolution -- much tighter than a blanket rtol."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from face_vijnana_yolov3_amd._lib import Context
    return Context(0)


def _ref_conv(x, w, k, s):
    """NHWC x, OHWI w, float64: ZeroPadding2D(1)+Conv2D valid (k=3) or 1x1."""
    xn = x.permute(0, 3, 1, 2)
    if k == 3:
        xn = F.pad(xn, (1, 1, 1, 1))
    return F.conv2d(xn, w.permute(0, 3, 1, 2), stride=s).permute(0, 2, 3, 1).contiguous()


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g, dtype=torch.float64) * (hi - lo) + lo).float()


def _check(got, ref, bound, what):
    err = (got.double().cpu() - ref).abs()
    tol = 2e-6 * bound + 1e-6
    bad = err > tol
    assert not bad.any(), '%s: max err %.3e, tol there %.3e, %d bad' % (what, err.max().item(), tol[bad].min().item() if bad.any() else 0, int(bad.sum()))


FWD_CASES = [
    # B, H, cin, cout, k, s
    (2, 16, 32, 64, 3, 2),
    (2, 16, 64, 32, 1, 1),
    (3, 13, 32, 64, 3, 1),      # M = 507: tail tile
    (2, 8, 128, 256, 3, 1),     # BN = 128 tiles
    (1, 26, 256, 128, 1, 1),
    (2, 12, 3, 32, 3, 1),       # first layer, gathered K = 27
    (2, 13, 1024, 6, 3, 1),     # head: N = 6 guard
    (1, 4, 512, 1024, 3, 2),
]


@pytest.mark.parametrize('B,H,cin,cout,k,s', FWD_CASES)
def test_conv_forward_raw_and_stats(ctx, B, H, cin, cout, k, s):
    from face_vijnana_yolov3_amd import ops
    x = _rand((B, H, H, cin), 1); w = _rand((cout, k, k, cin), 2)
    ref = _ref_conv(x.double(), w.double(), k, s)
    bound = _ref_conv(x.double().abs(), w.double().abs(), k, s)
    out, psum, psq = ops.conv2d_forward(ctx, x.cuda(), w.cuda(), stride=s, stats=True)
    _check(out, ref, bound, 'conv fwd')
    rows = ref.numel() // cout
    _check(psum.sum(0), ref.view(rows, cout).sum(0), bound.view(rows, cout).sum(0), 'psum')
    _check(psq.sum(0), (ref.view(rows, cout) ** 2).sum(0), (bound.view(rows, cout) ** 2).sum(0) * 2, 'psq')
    # the plain (no epilogue, no statistics) call: right by the same bound; and bit-identical to the statistics form whenever both run the
    # tile kernels (a small-M inference launch may take conv_small_kernel, which adds the K steps up in another order)
    out2 = ops.conv2d_forward(ctx, x.cuda(), w.cuda(), stride=s)
    _check(out2, ref, bound, 'conv fwd, plain call')
    ctx.set_option('conv_small', 0)
    try:
        out3 = ops.conv2d_forward(ctx, x.cuda(), w.cuda(), stride=s)
    finally:
        ctx.set_option('conv_small', 1)
    assert torch.equal(out, out3)


def test_conv_forward_fused_inference_epilogue(ctx):
    from face_vijnana_yolov3_amd import ops
    B, H, cin, cout = 2, 13, 64, 128
    x = _rand((B, H, H, cin), 3); w = _rand((cout, 3, 3, cin), 4)
    scale = _rand((cout,), 5, 0.5, 1.5); shift = _rand((cout,), 6); skip = _rand((B, H, H, cout), 7)
    ref = _ref_conv(x.double(), w.double(), 3, 1) * scale.double() + shift.double()
    ref = F.leaky_relu(ref, 0.1) + skip.double()
    bound = _ref_conv(x.double().abs(), w.double().abs(), 3, 1) * scale.double().abs() + 2.0
    out = ops.conv2d_forward(ctx, x.cuda(), w.cuda(), 1, scale.cuda(), shift.cuda(), 0.1, skip.cuda())
    _check(out, ref, bound, 'fused epilogue')
    # head form: bias only, linear
    out = ops.conv2d_forward(ctx, x.cuda(), w.cuda(), 1, None, shift.cuda(), -1.0, None)
    _check(out, _ref_conv(x.double(), w.double(), 3, 1) + shift.double(), bound, 'bias epilogue')


DGRAD_CASES = [
    (2, 16, 32, 64, 3, 1, 64),
    (2, 16, 32, 64, 3, 2, 64),
    (2, 13, 128, 64, 1, 1, 64),
    (3, 12, 64, 128, 3, 2, 128),
    (2, 13, 1024, 6, 3, 1, 32),   # head: dy padded to 32 channels
    (1, 26, 256, 512, 3, 2, 512),
]


@pytest.mark.parametrize('B,H,cin,cout,k,s,cpad', DGRAD_CASES)
def test_conv_dgrad(ctx, B, H, cin, cout, k, s, cpad):
    from face_vijnana_yolov3_amd import ops
    Ho = H // s
    w = _rand((cout, k, k, cin), 11)
    dy = torch.zeros((B, Ho, Ho, cpad)); dy[..., :cout] = _rand((B, Ho, Ho, cout), 12)
    add = _rand((B, H, H, cin), 13)
    x = torch.zeros((B, H, H, cin), dtype=torch.float64, requires_grad=True)
    y = _ref_conv(x, w.double(), k, s)
    (ref,) = torch.autograd.grad(y, x, dy[..., :cout].double())
    xa = torch.zeros((B, H, H, cin), dtype=torch.float64, requires_grad=True)
    (bound,) = torch.autograd.grad(_ref_conv(xa, w.double().abs(), k, s), xa, dy[..., :cout].double().abs())
    got = ops.conv2d_dgrad(ctx, dy.cuda(), w.cuda(), (H, H), s)
    _check(got, ref, bound, 'dgrad')
    got = ops.conv2d_dgrad(ctx, dy.cuda(), w.cuda(), (H, H), s, addend=add.cuda())
    _check(got, ref + add.double(), bound + 1.0, 'dgrad+add')


WGRAD_CASES = [
    (2, 16, 128, 128, 3, 1, 128),   # QUAD 128x128
    (2, 16, 64, 64, 3, 1, 64),
    (2, 16, 32, 64, 3, 2, 64),      # 64x32 tiles, stride 2
    (2, 16, 64, 32, 1, 1, 32),      # 32x64
    (3, 13, 32, 32, 3, 1, 32),      # 32x32, odd pixel count
    (2, 20, 3, 32, 3, 1, 32),       # first layer, gathered
    (2, 13, 1024, 6, 3, 1, 32),     # head
    (1, 26, 256, 128, 1, 1, 128),
    (4, 26, 128, 256, 3, 1, 256),
    (2, 16, 256, 512, 3, 1, 512),   # >= 64 (tile, tap) workgroups per K-split: plain split order
    # lattices smaller than one 32-pixel chunk: the row table's image / row / column carries all fire within a chunk
    (5, 3, 128, 128, 3, 1, 128),    # 3x3 lattice: a chunk spans 3.6 images
    (7, 4, 128, 256, 3, 2, 256),    # stride 2 -> 2x2 lattice: 8 images per chunk
    (3, 5, 64, 64, 3, 1, 64),       # split form, 25-pixel images
    (9, 2, 256, 128, 1, 1, 128),    # 1x1 on a 2x2 lattice
    (2, 33, 32, 64, 3, 1, 64),      # row length 33: the column wraps at a different lane every chunk
    # the fused nine-tap kernel (32 -> 64 channels): whole units, ragged units in both directions, both strides, padded dy
    (2, 16, 32, 64, 3, 1, 64),
    (3, 13, 32, 64, 3, 1, 64),
    (2, 20, 32, 64, 3, 2, 64),
    (1, 38, 32, 64, 3, 2, 128),
    (5, 4, 32, 64, 3, 1, 64),
    # the first layer's halo kernel: whole / ragged units, padded dy, more units than workgroups, one-row images
    (2, 32, 3, 32, 3, 1, 32),
    (3, 13, 3, 32, 3, 1, 32),
    (1, 70, 3, 32, 3, 1, 64),
    (200, 8, 3, 32, 3, 1, 32),
    (4, 1, 3, 32, 3, 1, 32),
    # the streaming 1x1 kernel (64 -> 32 channels): whole units, a ragged last unit, padded dy, fewer pixels than one unit
    (2, 16, 64, 32, 1, 1, 64),
    (3, 13, 64, 32, 1, 1, 32),
    (1, 5, 64, 32, 1, 1, 32),
    (9, 40, 64, 32, 1, 1, 32),
]


def test_first_layer_wgrad_halo_equals_gather(ctx):
    """wgrad0_mfma.hip against the element-wise gather kernel (option "wgrad_fused_taps" = 0): same products, other summation order."""
    from face_vijnana_yolov3_amd import ops
    for (B, H) in [(3, 48), (2, 21)]:
        x = _rand((B, H, H, 3), 83).cuda(); dy = _rand((B, H, H, 32), 84).cuda()
        got = ops.conv2d_wgrad(ctx, x, dy, 32, 3, 1)
        ctx.set_wgrad_fused_taps(False)
        try:
            ref = ops.conv2d_wgrad(ctx, x, dy, 32, 3, 1)
        finally:
            ctx.set_wgrad_fused_taps(True)
        assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item(), (B, H)


def test_wgrad_fused_taps_equals_generic(ctx):
    """option "wgrad_fused_taps": the nine-tap kernel and the one-workgroup-per-tap kernel form the same products; only the
    order of the float additions differs."""
    from face_vijnana_yolov3_amd import ops
    for (B, H, s) in [(3, 40, 1), (2, 52, 2)]:
        x = _rand((B, H, H, 32), 81).cuda(); dy = _rand((B, H // s, H // s, 64), 82).cuda()
        got = ops.conv2d_wgrad(ctx, x, dy, 64, 3, s)
        ctx.set_wgrad_fused_taps(False)
        try:
            ref = ops.conv2d_wgrad(ctx, x, dy, 64, 3, s)
        finally:
            ctx.set_wgrad_fused_taps(True)
        scale = ref.abs().max().item()
        assert (got - ref).abs().max().item() <= 2e-5 * scale, (B, H, s)


@pytest.mark.parametrize('B,H,cin,cout,k,s,ndy', WGRAD_CASES)
def test_conv_wgrad(ctx, B, H, cin, cout, k, s, ndy):
    from face_vijnana_yolov3_amd import ops
    Ho = H // s
    x = _rand((B, H, H, cin), 21)
    dy = torch.zeros((B, Ho, Ho, ndy)); dy[..., :cout] = _rand((B, Ho, Ho, cout), 22)
    w = torch.zeros((cout, k, k, cin), dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(_ref_conv(x.double(), w, k, s), w, dy[..., :cout].double())
    wa = torch.zeros((cout, k, k, cin), dtype=torch.float64, requires_grad=True)
    (bound,) = torch.autograd.grad(_ref_conv(x.double().abs(), wa, k, s), wa, dy[..., :cout].double().abs())
    got = ops.conv2d_wgrad(ctx, x.cuda(), dy.cuda(), cout, k, s)
    _check(got, ref, bound, 'wgrad')


@pytest.mark.parametrize('H', [140, 100])
def test_conv_tail_split_with_lent_scratch(ctx, H):
    """fv_set_conv_scratch, both plans of fv_conv_tail_plan: H = 100 -> 157 output tiles of 72 K steps, every tile cut into K
    slices; H = 140 -> 307 tiles, 256 stay whole (one per CU) and the other 51 are cut into slices that fill the second slot of
    the CUs.  The fix-up kernel applies the epilogue.  Forward (raw + BN partial sums, fused affine/leaky/add) and stride-1
    data-gradient against float64; the unsplit launch differs only in rounding."""
    from face_vijnana_yolov3_amd import ops
    B, cin, cout = 2, 256, 128
    x = _rand((B, H, H, cin), 31); w = _rand((cout, 3, 3, cin), 32, -0.1, 0.1)
    scale = _rand((cout,), 33, 0.5, 1.5); shift = _rand((cout,), 34); skip = _rand((B, H, H, cout), 35)
    ref = _ref_conv(x.double(), w.double(), 3, 1)
    bound = _ref_conv(x.double().abs(), w.double().abs(), 3, 1)
    xd, wd = x.cuda(), w.cuda()
    plain, _, _ = ops.conv2d_forward(ctx, xd, wd, stride=1, stats=True)
    ctx.set_conv_scratch(torch.empty(64 << 20, dtype=torch.uint8, device='cuda'))
    try:
        out, psum, psq = ops.conv2d_forward(ctx, xd, wd, stride=1, stats=True)
        out_again, _, _ = ops.conv2d_forward(ctx, xd, wd, stride=1, stats=True)
        fused = ops.conv2d_forward(ctx, xd, wd, 1, scale.cuda(), shift.cuda(), 0.1, skip.cuda())
        # data-gradient of a 128 -> 256 conv: the same gather problem with mirrored taps
        w2 = _rand((cin, 3, 3, cout), 36, -0.1, 0.1); dy = _rand((B, H, H, cin), 37)
        dgrad = ops.conv2d_dgrad(ctx, dy.cuda(), w2.cuda(), (H, H), 1)
    finally:
        ctx.set_conv_scratch(None)
    dgrad_plain = ops.conv2d_dgrad(ctx, dy.cuda(), w2.cuda(), (H, H), 1)
    assert torch.equal(out, out_again)                      # deterministic
    assert not torch.equal(out, plain)                      # the split really ran (other summation order)
    _check(out, ref, bound, 'tail-split fwd')
    _check(plain, ref, bound, 'unsplit fwd')
    rows = ref.numel() // cout
    _check(psum.sum(0), ref.view(rows, cout).sum(0), bound.view(rows, cout).sum(0), 'psum')
    _check(psq.sum(0), (ref.view(rows, cout) ** 2).sum(0), (bound.view(rows, cout) ** 2).sum(0) * 2, 'psq')
    ref_f = F.leaky_relu(ref * scale.double() + shift.double(), 0.1) + skip.double()
    _check(fused, ref_f, bound * scale.double().abs() + 2.0, 'tail-split fused epilogue')
    xg = torch.zeros((B, H, H, cout), dtype=torch.float64, requires_grad=True)
    (ref_d,) = torch.autograd.grad(_ref_conv(xg, w2.double(), 3, 1), xg, dy.double())
    xa = torch.zeros((B, H, H, cout), dtype=torch.float64, requires_grad=True)
    (bound_d,) = torch.autograd.grad(_ref_conv(xa, w2.double().abs(), 3, 1), xa, dy.double().abs())
    _check(dgrad, ref_d, bound_d, 'tail-split dgrad')
    assert not torch.equal(dgrad, dgrad_plain)


@pytest.mark.parametrize('B,H,cin,cout,k', [(4, 3, 512, 1024, 3), (4, 6, 512, 256, 1), (2, 13, 1024, 6, 3), (3, 8, 64, 128, 3),
                                            (2, 140, 256, 128, 3)])
def test_k_split_slabs_are_reused_safely(ctx, B, H, cin, cout, k):
    """Two different problems of one shape alternate through the SAME lent scratch: each result must be
    bit-identical every time (a slab line left over from the previous launch -- per-XCD L2s are not
    coherent -- would show up here) and within rounding of the launch without scratch.  The last shape
    takes the tail split (307 tiles); the small ones document that lending scratch is harmless there."""
    from face_vijnana_yolov3_amd import ops
    xs = [_rand((B, H, H, cin), 40 + i).cuda() for i in range(2)]
    ws = [_rand((cout, k, k, cin), 50 + i, -0.1, 0.1).cuda() for i in range(2)]
    sc = _rand((cout,), 60, 0.5, 1.5).cuda(); sh = _rand((cout,), 61).cuda()
    skip = _rand((B, H, H, cout), 62).cuda()
    plain = [ops.conv2d_forward(ctx, xs[i], ws[i], 1, sc, sh, 0.1, skip) for i in range(2)]
    ctx.set_conv_scratch(torch.empty(96 << 20, dtype=torch.uint8, device='cuda'))
    try:
        first = [None, None]
        for rep in range(6):
            i = rep & 1
            out = ops.conv2d_forward(ctx, xs[i], ws[i], 1, sc, sh, 0.1, skip)
            if first[i] is None:
                first[i] = out
            else:
                assert torch.equal(out, first[i]), (rep, (out - first[i]).abs().max().item())
    finally:
        ctx.set_conv_scratch(None)
    for i in range(2):
        d = (first[i] - plain[i]).abs().max().item()
        assert d <= 2e-5 * plain[i].abs().max().item(), d


@pytest.mark.parametrize('rows,C', [(1000, 32), (4097, 64), (338, 1024), (70000, 128)])
def test_bn_forward_backward(ctx, rows, C):
    from face_vijnana_yolov3_amd import ops
    from face_vijnana_yolov3_amd._lib import lib
    z = (_rand((rows, C), 31) * 2 + 0.3)
    gamma = _rand((C,), 32, 0.5, 1.5); beta = _rand((C,), 33); g = _rand((rows, C), 34); skip = _rand((rows, C), 35)
    mm = _rand((C,), 36); mv = _rand((C,), 37, 0.5, 2.0)
    # partials as the conv epilogue would emit them (128-row tiles)
    nt = lib().fv_conv2d_stat_rows(rows)
    zp = torch.zeros((nt * 128, C)); zp[:rows] = z
    psum = zp.view(nt, 128, C).sum(1); psq = (zp.view(nt, 128, C) ** 2).sum(1)
    mmd, mvd = mm.cuda(), mv.cuda()
    mean, invstd, scale, shift = ops.bn_finalize(ctx, psum.cuda(), psq.cuda(), rows, gamma.cuda(), beta.cuda(), 1e-3, 0.99, mmd, mvd)
    zd = z.double()
    rmean = zd.mean(0); rvar = zd.var(0, unbiased=False)
    torch.testing.assert_close(mean.cpu().double(), rmean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(invstd.cpu().double(), 1 / torch.sqrt(rvar + 1e-3), rtol=2e-5, atol=0)
    torch.testing.assert_close(mmd.cpu().double(), 0.99 * mm.double() + 0.01 * rmean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(mvd.cpu().double(), 0.99 * mv.double() + 0.01 * rvar * rows / (rows - 1.001), rtol=1e-5, atol=1e-6)
    # forward activation (+skip)
    zt = zd.clone().requires_grad_(True); gt = gamma.double().clone().requires_grad_(True); bt = beta.double().clone().requires_grad_(True)
    y = (zt - zt.mean(0)) / torch.sqrt(zt.var(0, unbiased=False) + 1e-3) * gt + bt
    a = F.leaky_relu(y, 0.1)
    out = ops.bn_act(ctx, z.cuda(), scale, shift, skip.cuda(), 0.1)
    torch.testing.assert_close(out.cpu().double(), (a + skip.double()).detach(), rtol=2e-5, atol=2e-5)
    # backward
    rdz, rdg, rdb = torch.autograd.grad(a, (zt, gt, bt), g.double())
    dz, dgamma, dbeta = ops.bn_bwd(ctx, g.cuda(), z.cuda(), scale, shift, mean, invstd, 0.1)
    sc = math.sqrt(rows)
    torch.testing.assert_close(dbeta.cpu().double(), rdb, rtol=1e-4, atol=1e-5 * sc)
    torch.testing.assert_close(dgamma.cpu().double(), rdg, rtol=1e-4, atol=1e-5 * sc)
    # elements whose pre-activation sits within float rounding of the LeakyReLU kink may pick the other slope
    near = (y.detach().abs() < 1e-5)
    err = (dz.cpu().double() - rdz).abs()
    assert (err[~near] <= 2e-5 + 1e-4 * rdz[~near].abs()).all(), err[~near].max()


def test_mse_loss_and_grad(ctx):
    from face_vijnana_yolov3_amd import ops
    yp = _rand((40 * 169, 6), 41, -2, 2); yt = _rand((40 * 169, 6), 42, 0, 1)
    loss, dy, db = ops.mse_loss_grad(ctx, yp.cuda(), yt.cuda(), 32)
    ref = ((yp.double() - yt.double()) ** 2).mean()
    assert abs(loss.item() - ref.item()) <= 1e-6 * ref.item()
    rdy = 2 * (yp.double() - yt.double()) / yp.numel()
    torch.testing.assert_close(dy.cpu()[:, :6].double(), rdy, rtol=1e-6, atol=1e-12)
    assert torch.count_nonzero(dy[:, 6:]) == 0
    torch.testing.assert_close(db.cpu().double(), rdy.sum(0), rtol=1e-5, atol=1e-9)


def test_adam_matches_keras_formula(ctx):
    from face_vijnana_yolov3_amd import ops
    from oracle import net_oracle as no
    n = 1000003  # not a multiple of 4: exercises the tail kernel
    p = _rand((n,), 51); g = _rand((n,), 52, -1e-2, 1e-2)
    m = torch.zeros(n); v = torch.zeros(n)
    pd, md, vd = p.cuda(), m.cuda(), v.cuda()
    rp, rm, rv = p.double(), m.double(), v.double()
    for it in range(3):
        ops.adam_step(ctx, pd, g.cuda(), md, vd, it, 1e-4, 0.99, 0.99, 1e-7, 0.01)
        rp, rm, rv = no.keras_adam(rp, g.double(), rm, rv, it, 1e-4, 0.99, 0.99, decay=0.01)
    torch.testing.assert_close(pd.cpu().double(), rp, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(md.cpu().double(), rm, rtol=1e-5, atol=1e-10)
    torch.testing.assert_close(vd.cpu().double(), rv, rtol=1e-5, atol=1e-12)


def test_fd_loss_and_grad(ctx):
    """fd_loss is defined but unused in the reference (fd.py:59-64); operator-level parity only."""
    from face_vijnana_yolov3_amd import ops
    from oracle import net_oracle as no
    yp = _rand((3 * 169, 6), 61, -0.3, 1.3)          # linear head: values outside [0,1] get clipped
    yt = (_rand((3 * 169, 6), 62, 0, 1) > 0.7).float(); yt[:, 1:5] = _rand((3 * 169, 4), 63, 0, 1)
    ypd = yp.double().clone().requires_grad_(True)
    ref = no.fd_loss(ypd, yt.double())
    (rg,) = torch.autograd.grad(ref, ypd)
    loss, dy = ops.fd_loss_grad(ctx, yp.cuda(), yt.cuda(), 32)
    assert abs(loss.item() - ref.item()) <= 2e-6 * abs(ref.item())
    torch.testing.assert_close(dy.cpu()[:, :6].double(), rg, rtol=2e-6, atol=1e-9)
    assert torch.count_nonzero(dy[:, 6:]) == 0


@pytest.mark.parametrize('B,H,cin,cout,k,s', [(2, 13, 128, 256, 3, 1), (2, 16, 32, 64, 3, 2), (3, 13, 64, 32, 1, 1), (2, 13, 1024, 6, 3, 1)])
def test_four_and_eight_wave_tiles_are_bit_identical(ctx, B, H, cin, cout, k, s):
    """option "conv_waves8": 2x4 waves of 64x32 (default) against 2x2 waves of 64x64 -- the same k-ordered fmaf chain per output
    element, so outputs (incl. zero padding at the borders) and data-gradients are bit-identical."""
    from face_vijnana_yolov3_amd import ops
    x = _rand((B, H, H, cin), 71).cuda(); w = _rand((cout, k, k, cin), 72).cuda()
    dy = _rand((B, H // s, H // s, max(32, cout)), 73).cuda()
    if cout < 32:
        dy[..., cout:] = 0
    ref = ops.conv2d_forward(ctx, x, w, stride=s)
    dg_ref = ops.conv2d_dgrad(ctx, dy, w, (H, H), s)
    ctx.set_conv_waves8(False)
    try:
        got = ops.conv2d_forward(ctx, x, w, stride=s)
        dg = ops.conv2d_dgrad(ctx, dy, w, (H, H), s)
    finally:
        ctx.set_conv_waves8(True)
    assert torch.equal(got, ref) and torch.equal(dg, dg_ref)
