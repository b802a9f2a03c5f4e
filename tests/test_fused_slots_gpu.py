"""Operator-level GPU parity of the FUSED forms fv_train_step runs (they used to be reachable only through
the whole step): conv forward -> fp64 accumulator slots -> bn_act_stats; data-gradient with the fused
BatchNorm-backward reduction (FV_EPI_BNRED: stride 1, stride-2 four-class launches, residual addend, and
the tail-split fix-up variant) -> bn_bwd_apply_slots.  Reference: float64 torch-CPU restatement of the
same Keras ops (yolov3_detect.py:212-215 BatchNormalization(eps 1e-3) + LeakyReLU(0.1); their gradients).

Tolerances are a few fp32 ulps of the absolute-value sums (the `_check` bound of test_ops_gpu.py).  The
LeakyReLU mask of an element is decided by the sign of fl(fl(z*scale)+shift) -- the library is built with
-ffp-contract=off, so torch's float32 `z*scale+shift` reproduces that decision bit for bit and no element
has to be excluded around the kink."""
import math

import numpy as np

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

LEAKY = 0.1


@pytest.fixture(scope='module')
def ctx():
    from face_vijnana_yolov3_amd._lib import Context
    return Context(0)


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g, dtype=torch.float64) * (hi - lo) + lo).float()


def _ref_conv(x, w, k, s):
    xn = x.permute(0, 3, 1, 2)
    if k == 3:
        xn = F.pad(xn, (1, 1, 1, 1))
    return F.conv2d(xn, w.permute(0, 3, 1, 2), stride=s).permute(0, 2, 3, 1).contiguous()


def _dgrad_ref(dy, w, B, H, cin, k, s):
    """float64 data-gradient and its absolute-value bound."""
    x = torch.zeros((B, H, H, cin), dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(_ref_conv(x, w.double(), k, s), x, dy.double())
    xa = torch.zeros((B, H, H, cin), dtype=torch.float64, requires_grad=True)
    (bound,) = torch.autograd.grad(_ref_conv(xa, w.double().abs(), k, s), xa, dy.double().abs())
    return ref, bound


def _bn_vectors(C, seed):
    """Arbitrary (not mutually consistent) per-channel vectors: the kernels take them as inputs."""
    return (_rand((C,), seed, 0.5, 1.5), _rand((C,), seed + 1, -0.5, 0.5), _rand((C,), seed + 2, -0.3, 0.3), _rand((C,), seed + 3, 0.5, 2.0))


def _mask(z, scale, shift):
    """The kernels' slope decision, bit for bit: fl(fl(z*scale)+shift) > 0 in float32."""
    pre = z * scale + shift
    return torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, LEAKY)).double()


BNRED_CASES = [
    # B, H, cin, cout, k, s, cpad, addend
    (2, 16, 64, 128, 3, 1, 128, False),     # 64-wide tiles
    (2, 16, 128, 256, 3, 1, 256, True),     # 128-wide tiles + residual addend
    (2, 16, 32, 64, 3, 2, 64, False),       # stride 2: four parity classes, 32-wide tiles
    (3, 12, 128, 128, 3, 2, 128, False),    # stride 2, 128-wide tiles, M = 108 per class (tail rows)
    (2, 13, 256, 128, 1, 1, 128, True),     # 1x1, odd pixel count
    (2, 13, 1024, 6, 3, 1, 32, False),      # the head's data-gradient (dy padded to 32 channels) -> conv_73's BN
]


def _run_bnred(ctx, B, H, cin, cout, k, s, cpad, with_add, seed=100):
    from face_vijnana_yolov3_amd import ops
    Ho = H // s
    w = _rand((cout, k, k, cin), seed, -0.3, 0.3)
    dy = torch.zeros((B, Ho, Ho, cpad)); dy[..., :cout] = _rand((B, Ho, Ho, cout), seed + 1)
    add = _rand((B, H, H, cin), seed + 2) if with_add else None
    z = _rand((B, H, H, cin), seed + 3, -2.0, 2.0)
    scale, shift, mean, invstd = _bn_vectors(cin, seed + 4)
    slots = ops.stat_slots(cin, 'cuda')
    dx = ops.conv2d_dgrad_bnred(ctx, dy.cuda(), w.cuda(), (H, H), s, z.cuda(), scale.cuda(), shift.cuda(), mean.cuda(), invstd.cuda(),
                                slots, addend=None if add is None else add.cuda())
    ref, bound = _dgrad_ref(dy[..., :cout], w, B, H, cin, k, s)
    if add is not None:
        ref = ref + add.double(); bound = bound + add.double().abs()
    return dx, slots, ref, bound, z, (scale, shift, mean, invstd)


@pytest.mark.parametrize('B,H,cin,cout,k,s,cpad,with_add', BNRED_CASES)
def test_dgrad_with_fused_bn_backward_reduction(ctx, B, H, cin, cout, k, s, cpad, with_add):
    dx, slots, ref, bound, z, (scale, shift, mean, invstd) = _run_bnred(ctx, B, H, cin, cout, k, s, cpad, with_add)
    err = (dx.double().cpu() - ref).abs()
    assert (err <= 2e-6 * bound + 1e-6).all(), 'dgrad: max err %.3e' % err.max().item()
    m = _mask(z, scale, shift)
    xhat = ((z - mean) * invstd).double()          # the kernel's float32 xhat, exactly
    rows = ref.numel() // cin
    gy = (ref * m).view(rows, cin)
    s_ = slots.sum(0).cpu()                        # [2][C] float64
    for name, got, want, mag in (('d-beta', s_[0], gy.sum(0), (bound * m).view(rows, cin).sum(0)),
                                 ('d-gamma', s_[1], (gy * xhat.view(rows, cin)).sum(0), (bound * m * xhat.abs()).view(rows, cin).sum(0))):
        e = (got - want).abs()
        tol = 1e-5 * mag + 1e-6
        assert (e <= tol).all(), '%s: max err %.3e (tol there %.3e, |want| max %.3e)' % (name, e.max().item(), tol[e.argmax()].item(), want.abs().max().item())
    # every tile landed in some slot and nothing else was touched
    assert torch.isfinite(slots).all()


def test_dgrad_bnred_through_the_tail_split_fixup(ctx):
    """307 output tiles x 72 K steps: with scratch lent the launcher cuts the tail tiles into K slices and
    conv_tail_fixup_kernel performs the epilogue -- including the fused BN-backward reduction."""
    from face_vijnana_yolov3_amd import ops
    B, H, cin, cout, k, s = 2, 140, 128, 256, 3, 1
    plain = _run_bnred(ctx, B, H, cin, cout, k, s, cout, True, seed=300)
    ctx.set_conv_scratch(torch.empty(64 << 20, dtype=torch.uint8, device='cuda'))
    try:
        split = _run_bnred(ctx, B, H, cin, cout, k, s, cout, True, seed=300)
    finally:
        ctx.set_conv_scratch(None)
    assert not torch.equal(plain[0], split[0])      # the split really ran (other fp32 summation order)
    for dx, slots, ref, bound, z, (scale, shift, mean, invstd) in (plain, split):
        err = (dx.double().cpu() - ref).abs()
        assert (err <= 2e-6 * bound + 1e-6).all(), err.max().item()
        m = _mask(z, scale, shift); xhat = ((z - mean) * invstd).double()
        rows = ref.numel() // cin
        gy = (ref * m).view(rows, cin)
        s_ = slots.sum(0).cpu()
        assert ((s_[0] - gy.sum(0)).abs() <= 1e-5 * (bound * m).view(rows, cin).sum(0) + 1e-6).all()
        assert ((s_[1] - (gy * xhat.view(rows, cin)).sum(0)).abs() <= 1e-5 * (bound * m * xhat.abs()).view(rows, cin).sum(0) + 1e-6).all()


@pytest.mark.parametrize('rows,C,reduced', [(1000, 32, False), (4097, 64, False), (338, 1024, False), (70000, 128, False),
                                            (6760, 512, True), (27040, 256, True)])
def test_bn_backward_apply_from_slots(ctx, rows, C, reduced):
    """bn_bwd_apply_slots_kernel: dz, d-beta, d-gamma from the slots -- filled by the kernel's own reduction
    pass (reduced=False) or beforehand, the way the data-gradient epilogue leaves them (reduced=True: exact
    per-128-row column sums spread over the slots)."""
    from face_vijnana_yolov3_amd import ops
    z = _rand((rows, C), 41, -2.0, 2.0); g = _rand((rows, C), 42)
    scale, shift, mean, invstd = _bn_vectors(C, 43)
    m = _mask(z, scale, shift); xhat = ((z - mean) * invstd).double()
    gy = g.double() * m
    slots = ops.stat_slots(C, 'cuda')
    if reduced:
        ns = slots.shape[0]
        nt = (rows + 127) // 128
        pad = torch.zeros((nt * 128, C), dtype=torch.float64)
        a = pad.clone(); a[:rows] = gy; b = pad.clone(); b[:rows] = gy * xhat
        a = a.view(nt, 128, C).sum(1); b = b.view(nt, 128, C).sum(1)
        host = torch.zeros((ns, 2, C), dtype=torch.float64)
        for t in range(nt):
            host[t % ns, 0] += a[t]; host[t % ns, 1] += b[t]
        slots.copy_(host)
    dz, dgamma, dbeta = ops.bn_bwd_slots(ctx, g.cuda(), z.cuda(), scale.cuda(), shift.cuda(), mean.cuda(), invstd.cuda(), slots, reduced)
    rdb, rdg = gy.sum(0), (gy * xhat).sum(0)
    mag_b, mag_g = gy.abs().sum(0), (gy * xhat).abs().sum(0)
    assert ((dbeta.cpu().double() - rdb).abs() <= 1e-5 * mag_b + 1e-6).all()
    assert ((dgamma.cpu().double() - rdg).abs() <= 1e-5 * mag_g + 1e-6).all()
    # dz from the kernel's own (float) d-beta / d-gamma: isolates the apply arithmetic
    db, dg = dbeta.cpu().double(), dgamma.cpu().double()
    rdz = scale.double() * (gy - db / rows - xhat * (dg / rows))
    mag = scale.double().abs() * (gy.abs() + db.abs() / rows + xhat.abs() * dg.abs() / rows)
    err = (dz.cpu().double() - rdz).abs()
    assert (err <= 1e-6 * mag + 1e-7).all(), err.max().item()
    # and the textbook gradient (autograd through batch statistics) when the vectors ARE the batch statistics
    if not reduced:
        zt = z.double().clone().requires_grad_(True)
        gam = _rand((C,), 47, 0.5, 1.5).double().requires_grad_(True); bet = _rand((C,), 48).double().requires_grad_(True)
        mu = zt.mean(0); var = zt.var(0, unbiased=False)
        y = (zt - mu) / torch.sqrt(var + 1e-3) * gam + bet
        act = F.leaky_relu(y, LEAKY)
        adz, adg, adb = torch.autograd.grad(act, (zt, gam, bet), g.double())
        is32 = (1.0 / torch.sqrt(var.detach() + 1e-3)).float(); mu32 = mu.detach().float()
        sc32 = gam.detach().float() * is32; sh32 = bet.detach().float() - mu32 * sc32
        slots.zero_()
        dz2, dg2, db2 = ops.bn_bwd_slots(ctx, g.cuda(), z.cuda(), sc32.cuda(), sh32.cuda(), mu32.cuda(), is32.cuda(), slots, False)
        near = (y.detach().abs() < 1e-5)            # float32 statistics vs float64 ones can disagree about the slope there
        sc = math.sqrt(rows)
        torch.testing.assert_close(db2.cpu().double(), adb, rtol=1e-4, atol=1e-5 * sc)
        torch.testing.assert_close(dg2.cpu().double(), adg, rtol=1e-4, atol=1e-5 * sc)
        e2 = (dz2.cpu().double() - adz).abs()
        assert (e2[~near] <= 2e-5 + 1e-4 * adz[~near].abs()).all(), e2[~near].max().item()


@pytest.mark.parametrize('B,H,cin,cout,k,s,with_skip', [(2, 16, 32, 64, 3, 2, False), (3, 13, 64, 128, 3, 1, True), (1, 26, 256, 128, 1, 1, False),
                                                         (2, 12, 3, 32, 3, 1, False), (2, 64, 3, 32, 3, 1, False), (4, 13, 512, 1024, 3, 1, True)])
def test_conv_slots_then_bn_act(ctx, B, H, cin, cout, k, s, with_skip):
    """conv forward adding its column sums to the slots, then bn_act_stats_kernel: statistics, moving
    statistics (Keras: var * n/(n-(1+eps)), momentum 0.99), scale/shift and the activated output."""
    from face_vijnana_yolov3_amd import ops
    x = _rand((B, H, H, cin), 51); w = _rand((cout, k, k, cin), 52, -0.2, 0.2)
    gamma = _rand((cout,), 53, 0.5, 1.5); beta = _rand((cout,), 54)
    mm = _rand((cout,), 55); mv = _rand((cout,), 56, 0.5, 2.0)
    Ho = H // s
    skip = _rand((B, Ho, Ho, cout), 57) if with_skip else None
    slots = ops.stat_slots(cout, 'cuda')
    z = ops.conv2d_forward_slots(ctx, x.cuda(), w.cuda(), s, slots)
    ref = _ref_conv(x.double(), w.double(), k, s); bound = _ref_conv(x.double().abs(), w.double().abs(), k, s)
    assert ((z.double().cpu() - ref).abs() <= 2e-6 * bound + 1e-6).all()
    rows = ref.numel() // cout
    zc = z.cpu().double().view(rows, cout)
    s_ = slots.sum(0).cpu()
    # column sums of the kernel's own z: fp32 per-tile partials, fp64 across tiles
    assert ((s_[0] - zc.sum(0)).abs() <= 1e-5 * zc.abs().sum(0) + 1e-6).all()
    assert ((s_[1] - (zc * zc).sum(0)).abs() <= 1e-5 * (zc * zc).sum(0) + 1e-6).all()
    mmd, mvd = mm.cuda(), mv.cuda()
    out, mean, invstd, scale, shift = ops.bn_act_slots(ctx, z, slots, gamma.cuda(), beta.cuda(), 1e-3, 0.99, mmd, mvd,
                                                       None if skip is None else skip.cuda())
    rmean = zc.mean(0); rvar = zc.var(0, unbiased=False)
    torch.testing.assert_close(mean.cpu().double(), rmean, rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(invstd.cpu().double(), 1 / torch.sqrt(rvar + 1e-3), rtol=2e-5, atol=0)
    torch.testing.assert_close(mmd.cpu().double(), 0.99 * mm.double() + 0.01 * rmean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(mvd.cpu().double(), 0.99 * mv.double() + 0.01 * rvar * rows / (rows - 1.001), rtol=1e-5, atol=1e-6)
    # activation from the kernel's own published scale / shift: exact arithmetic check
    sc, sh = scale.cpu(), shift.cpu()
    torch.testing.assert_close(sc.double(), gamma.double() * invstd.cpu().double(), rtol=1e-6, atol=0)
    pre = z.cpu().view(rows, cout) * sc + sh
    want = torch.where(pre > 0, pre, pre * LEAKY)
    if skip is not None:
        want = want + skip.view(rows, cout)
    assert torch.equal(out.cpu().view(rows, cout), want)


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B,H,W', [(2, 64, 64), (1, 96, 128), (3, 8, 32)])
def test_first_layer_direct_kernel_equals_gather_kernel(ctx, B, H, W):
    """conv0_direct.hip (vector-FMA first layer with an LDS halo tile; taken when W % 32 == 0 and H % 8 == 0) against
    the matrix-core gather kernel: the same k-ordered fmaf chain, so raw outputs and the fused inference epilogue are
    bit-identical; the training statistics (other fp32 partial-sum order) agree to rounding and with float64."""
    from face_vijnana_yolov3_amd import ops
    from face_vijnana_yolov3_amd._lib import lib, ptr
    x = _rand((B, H, W, 3), 91, 0.0, 1.0).cuda(); w = _rand((32, 3, 3, 3), 92, -0.5, 0.5).cuda()
    scale = _rand((32,), 93, 0.5, 1.5).cuda(); shift = _rand((32,), 94).cuda()
    res = {}
    try:
        for direct in (True, False):
            ctx.set_conv0_direct(direct)
            slots = ops.stat_slots(32, 'cuda')
            wd = ops.pack_first_layer(ctx, w)
            z = torch.empty((B, H, W, 32), dtype=torch.float32, device='cuda')
            ctx.check(lib().fv_conv2d_forward_slots(ctx.handle, ptr(x), ptr(wd), B, H, W, 3, 32, 3, 1, ptr(z), ptr(slots), slots.shape[0]), 'slots')
            raw = torch.empty_like(z); fused = torch.empty_like(z)
            NULL = ops.NULL
            ctx.check(lib().fv_conv2d_forward(ctx.handle, ptr(x), ptr(wd), B, H, W, 3, 32, 3, 1, NULL, NULL, -1.0, NULL, ptr(raw), NULL, NULL), 'raw')
            ctx.check(lib().fv_conv2d_forward(ctx.handle, ptr(x), ptr(wd), B, H, W, 3, 32, 3, 1, ptr(scale), ptr(shift), 0.1, NULL, ptr(fused), NULL, NULL), 'fused')
            res[direct] = (z, slots.sum(0), raw, fused)
    finally:
        ctx.set_conv0_direct(True)
    for k in (0, 2, 3):
        assert torch.equal(res[True][k], res[False][k]), k
    zc = res[True][0].cpu().double().view(-1, 32)
    for s_ in (res[True][1].cpu(), res[False][1].cpu()):
        assert ((s_[0] - zc.sum(0)).abs() <= 1e-5 * zc.abs().sum(0) + 1e-6).all()
        assert ((s_[1] - (zc * zc).sum(0)).abs() <= 1e-5 * (zc * zc).sum(0) + 1e-6).all()
    ref = _ref_conv(x.cpu().double(), w.cpu().double(), 3, 1); bound = _ref_conv(x.cpu().double().abs(), w.cpu().double().abs(), 3, 1)
    assert ((res[True][2].cpu().double() - ref).abs() <= 2e-6 * bound + 1e-6).all()


@pytest.mark.parametrize('B,H,cin,cout,k,s', [(2, 16, 128, 256, 3, 1), (3, 12, 128, 128, 3, 2), (2, 13, 256, 128, 1, 1), (2, 140, 256, 128, 3, 1),
                                            (2, 16, 64, 128, 3, 1), (2, 16, 32, 64, 3, 2), (3, 13, 64, 64, 1, 1)])
def test_eight_wave_conv_equals_four_wave_conv(ctx, B, H, cin, cout, k, s):
    """option "conv_waves8": 512-thread workgroups (8 waves of 64x32) against 256-thread ones (4 waves of 64x64) on the
    128- and 64-wide tiles.  Every output element is the same k-ordered fmaf chain: forward (raw, fused epilogue, BN partial
    sums) and data-gradient are bit-identical; the fused BN-backward column sums agree to float32 rounding (the two
    row lanes of a wave are pre-added by a shuffle in the 8-wave form).  The last shape takes the tail split."""
    from face_vijnana_yolov3_amd import ops
    x = _rand((B, H, H, cin), 111).cuda(); w = _rand((cout, k, k, cin), 112, -0.2, 0.2).cuda()
    Ho = H // s
    dy = _rand((B, Ho, Ho, cout), 113).cuda(); add = _rand((B, H, H, cin), 114).cuda()
    z2 = _rand((B, H, H, cin), 115, -2.0, 2.0).cuda(); v2 = [v.cuda() for v in _bn_vectors(cin, 116)]
    scale = _rand((cout,), 117, 0.5, 1.5).cuda(); shift = _rand((cout,), 118).cuda()
    res = {}
    if H == 140:
        ctx.set_conv_scratch(torch.empty(64 << 20, dtype=torch.uint8, device='cuda'))
    try:
        for w8 in (True, False):
            ctx.set_conv_waves8(w8)
            out, psum, psq = ops.conv2d_forward(ctx, x, w, stride=s, stats=True)
            fused = ops.conv2d_forward(ctx, x, w, s, scale, shift, 0.1, None)
            slots = ops.stat_slots(cin, 'cuda')
            dx = ops.conv2d_dgrad_bnred(ctx, dy, w, (H, H), s, z2, *v2, slots, addend=add)
            res[w8] = (out, psum, psq, fused, dx, slots.sum(0))
    finally:
        ctx.set_conv_waves8(True); ctx.set_conv_scratch(None)
    for i in range(5):
        if i in (1, 2) and cout <= 64:
            # 64-wide tiles: the 8-wave form is 4 x 2 waves (32 rows each) against 2 x 2 (64 rows each) -- the per-tile column
            # sums are added up in another order; equal to float32 rounding of the sums
            torch.testing.assert_close(res[True][i].sum(0), res[False][i].sum(0), rtol=2e-5, atol=2e-5 * res[False][i].sum(0).abs().max().item())
            continue
        assert torch.equal(res[True][i], res[False][i]), i
    mag = res[False][5].abs().max().item()
    torch.testing.assert_close(res[True][5], res[False][5], rtol=2e-5, atol=2e-5 * mag)


@pytest.mark.parametrize('B,H,s', [(2, 16, 1), (3, 13, 1), (2, 20, 2), (1, 38, 2), (5, 4, 1), (2, 48, 2), (41, 16, 1)])
def test_halo_forward_is_bit_identical_to_the_tile_kernel(ctx, B, H, s):
    """option "conv_halo" (conv9_mfma.hip: 32 -> 64 channels, 3x3, training forward): resident weights + one x halo
    tile per 8x16-pixel unit, same k-ordered fmaf chain as conv_kernel<64,...> -> z bit-identical (whole, ragged and tiny
    units, both strides, more units than workgroups); the statistics slots hold the same column sums in another order."""
    from face_vijnana_yolov3_amd import ops
    x = _rand((B, H, H, 32), 91).cuda(); w = _rand((64, 3, 3, 32), 92, -0.2, 0.2).cuda()
    sl = ops.stat_slots(64, 'cuda')
    z = ops.conv2d_forward_slots(ctx, x, w, s, sl)
    ctx.set_conv_halo(False)
    try:
        sl0 = ops.stat_slots(64, 'cuda')
        z0 = ops.conv2d_forward_slots(ctx, x, w, s, sl0)
    finally:
        ctx.set_conv_halo(True)
    assert torch.equal(z, z0)
    zc = z.double().view(-1, 64)
    s1, s0 = sl.sum(0).cpu(), sl0.sum(0).cpu()
    assert ((s1[0] - zc.sum(0).cpu()).abs() <= 1e-6 * zc.abs().sum(0).cpu() + 1e-9).all()
    assert ((s1[1] - (zc * zc).sum(0).cpu()).abs() <= 1e-6 * (zc * zc).sum(0).cpu() + 1e-9).all()
    assert ((s1 - s0).abs() <= 2e-5 * s0.abs() + 1e-5).all()


@pytest.mark.parametrize('B,H,ndy,bn', [(2, 32, 64, True), (3, 26, 64, True), (1, 76, 64, False), (2, 16, 64, True), (41, 32, 64, True), (2, 44, 64, False)])
def test_halo_stride2_dgrad_is_bit_identical_to_the_tile_kernel(ctx, B, H, ndy, bn):
    """option "conv_halo", data-gradient side (dgrad9s2_mfma.hip: the stride-2 3x3 layer with 32 -> 64 channels): the four
    parity classes from one dy halo tile with the transposed weights resident in LDS; same k-ordered chains as
    conv_kernel<32,4,1> -> dx bit-identical (whole and ragged units, more units than workgroups); with the fused
    BN-backward reduction the slots hold the same d-beta / d-gamma sums in another order, and match float64."""
    from face_vijnana_yolov3_amd import ops
    Ho = H // 2
    w = _rand((64, 3, 3, 32), 101, -0.2, 0.2).cuda(); dy = _rand((B, Ho, Ho, ndy), 102).cuda()
    z = (_rand((B, H, H, 32), 103) * 2 + 0.2).cuda()
    gamma = _rand((32,), 104, 0.5, 1.5); beta = _rand((32,), 105)
    zc = z.double().view(-1, 32)
    mean = zc.mean(0); invstd = 1 / torch.sqrt(zc.var(0, unbiased=False) + 1e-3)
    scale = (gamma.cuda().double() * invstd).float(); shift = (beta.cuda().double() - mean * scale.double()).float()
    meanf, invf = mean.float(), invstd.float()

    def run():
        if not bn:
            return ops.conv2d_dgrad(ctx, dy, w, (H, H), 2), None
        sl = ops.stat_slots(32, 'cuda')
        return ops.conv2d_dgrad_bnred(ctx, dy, w, (H, H), 2, z, scale, shift, meanf, invf, sl), sl
    dx, sl = run()
    ctx.set_conv_halo(False)
    try:
        dx0, sl0 = run()
    finally:
        ctx.set_conv_halo(True)
    assert torch.equal(dx, dx0)
    if bn:
        s1, s0 = sl.sum(0).cpu(), sl0.sum(0).cpu()
        g = dx.double().view(-1, 32)
        gy = torch.where((z.view(-1, 32) * scale + shift) > 0, g, g * 0.1)          # the device's own branch decision
        xhat = (z.view(-1, 32).double() - meanf.double()) * invf.double()
        ref = torch.stack([gy.sum(0), (gy * xhat).sum(0)]).cpu()
        mag = torch.stack([gy.abs().sum(0), (gy * xhat).abs().sum(0)]).cpu()
        assert ((s1 - ref).abs() <= 2e-6 * mag + 1e-9).all(), (s1 - ref).abs().max().item()
        assert ((s1 - s0).abs() <= 2e-5 * mag + 1e-6).all()


@pytest.mark.parametrize('B', [14500, 14600])
def test_halo_forward_next_to_the_2_gib_output_bound(ctx, B):
    """conv9_mfma.hip masks edge stores with the 32-bit byte offset 0x80000000, which must lie beyond the output: the halo kernel
    takes outputs below 2 GiB (B = 14500 at 24x24x64: 2.138e9 bytes, partial 8x16 units at the bottom and right edges) and leaves
    larger ones (B = 14600: 2.153e9) to the tile kernel.  Either way the first and last images must be bit-identical to the same
    images convolved on their own (before the guard, masked stores of the larger case landed inside the tensor)."""
    from face_vijnana_yolov3_amd import ops
    H = 24
    g = torch.Generator(device='cuda').manual_seed(7)
    x = torch.rand((B, H, H, 32), generator=g, device='cuda') * 2 - 1
    w = _rand((64, 3, 3, 32), 92, -0.2, 0.2).cuda()
    assert (B * H * H * 64 * 4 >= 2 ** 31) == (B == 14600)
    z = ops.conv2d_forward_slots(ctx, x, w, 1, ops.stat_slots(64, 'cuda'))
    for sl in (slice(0, 3), slice(B // 2, B // 2 + 3), slice(B - 3, B)):
        part = ops.conv2d_forward_slots(ctx, x[sl].contiguous(), w, 1, ops.stat_slots(64, 'cuda'))
        assert torch.equal(z[sl], part)
    ctx.set_conv_halo(False)
    try:
        part0 = ops.conv2d_forward_slots(ctx, x[B - 3:].contiguous(), w, 1, ops.stat_slots(64, 'cuda'))
    finally:
        ctx.set_conv_halo(True)
    assert torch.equal(z[B - 3:], part0)


@pytest.mark.parametrize('B', [8600, 8700])
def test_halo_stride2_dgrad_next_to_the_2_gib_output_bound(ctx, B):
    """Same bound for dgrad9s2_mfma.hip (dx of the 32 -> 64 stride-2 layer, 44x44x32 per image: 2.131e9 bytes at B = 8600,
    2.156e9 at B = 8700; 44 is not a multiple of the unit, so edge units are partial)."""
    from face_vijnana_yolov3_amd import ops
    H = 44
    g = torch.Generator(device='cuda').manual_seed(8)
    dy = torch.rand((B, H // 2, H // 2, 64), generator=g, device='cuda') * 2 - 1
    w = _rand((64, 3, 3, 32), 101, -0.2, 0.2).cuda()
    assert (B * H * H * 32 * 4 >= 2 ** 31) == (B == 8700)
    dx = ops.conv2d_dgrad(ctx, dy, w, (H, H), 2)
    for sl in (slice(0, 3), slice(B - 3, B)):
        assert torch.equal(dx[sl], ops.conv2d_dgrad(ctx, dy[sl].contiguous(), w, (H, H), 2))
    ctx.set_conv_halo(False)
    try:
        part0 = ops.conv2d_dgrad(ctx, dy[B - 3:].contiguous(), w, (H, H), 2)
    finally:
        ctx.set_conv_halo(True)
    assert torch.equal(dx[B - 3:], part0)


# rows (B x H x H), cin, cout: every shape has more than 512 tiles, so the persistent form is taken and workgroups walk 2+ tiles
PERSIST_CASES = [
    (7, 100, 256, 128),     # 547 tiles of 128x128, 8 K steps, ragged last M tile (70 000 rows)
    (5, 84, 128, 256),      # 276 x 2 tiles, 4 K steps: two N tiles per A panel
    (5, 84, 96, 256),       # 3 K steps: odd step count (the loop is unrolled by two)
    (9, 96, 32, 128),       # ONE K step per tile, 648 tiles
    (9, 96, 64, 128),       # two K steps
    (9, 92, 128, 64),       # 64-wide tiles (4 x 2 waves), 596 tiles
    (3, 64, 64, 1024),      # 96 x 8 tiles: eight N tiles per A panel
]


@pytest.mark.parametrize('B,H,cin,cout', PERSIST_CASES)
def test_persistent_1x1_conv_equals_the_tile_kernel(ctx, B, H, cin, cout):
    """option "conv1x1_persist" (conv1x1_mfma.hip): 1x1 launches with more than 512 tiles as a persistent GEMM whose workgroups
    walk several tiles.  Same tile shape, K order and MFMA sequence per output element as conv_kernel: the training forward (raw
    z + per-tile partial sums), the inference epilogue (affine + LeakyReLU + residual) and the data-gradient with addend and fused
    BN-backward reduction are BIT-IDENTICAL to the one-tile-per-workgroup launches; the fp64 slot sums hold the same per-tile
    column sums, added in another order.  And the result is right (float64 reference)."""
    from face_vijnana_yolov3_amd import ops
    x = _rand((B, H, H, cin), 211).cuda(); w = _rand((cout, 1, 1, cin), 212, -0.2, 0.2).cuda()
    dy = _rand((B, H, H, cout), 213).cuda(); add_in = _rand((B, H, H, cin), 214).cuda(); add_out = _rand((B, H, H, cout), 219).cuda()
    z2 = _rand((B, H, H, cin), 215, -2.0, 2.0).cuda(); v2 = [v.cuda() for v in _bn_vectors(cin, 216)]
    scale = _rand((cout,), 217, 0.5, 1.5).cuda(); shift = _rand((cout,), 218).cuda()
    res = {}
    assert ctx.get_option('conv1x1_persist') == 1          # the default
    try:
        for on in (True, False):
            ctx.set_option('conv1x1_persist', on)
            out, psum, psq = ops.conv2d_forward(ctx, x, w, stride=1, stats=True)
            fslots = ops.stat_slots(cout, 'cuda')
            zs = ops.conv2d_forward_slots(ctx, x, w, 1, fslots)
            fused = ops.conv2d_forward(ctx, x, w, 1, scale, shift, 0.1, add_out)
            plain = ops.conv2d_dgrad(ctx, dy, w, (H, H), 1)
            bslots = ops.stat_slots(cin, 'cuda')
            dx = ops.conv2d_dgrad_bnred(ctx, dy, w, (H, H), 1, z2, *v2, bslots, addend=add_in) if (256 % cin == 0 or cin % 256 == 0) else plain
            torch.cuda.synchronize()
            res[on] = (out, psum, psq, zs, fused, plain, dx, fslots.sum(0), bslots.sum(0))
    finally:
        ctx.set_option('conv1x1_persist', 1)
    for i in range(7):
        assert torch.equal(res[True][i], res[False][i]), i
    for i in (7, 8):
        mag = res[False][i].abs().max().item()
        torch.testing.assert_close(res[True][i], res[False][i], rtol=1e-12, atol=1e-12 * max(mag, 1.0))
    ref = _ref_conv(x.cpu().double(), w.cpu().double(), 1, 1); bound = _ref_conv(x.cpu().double().abs(), w.cpu().double().abs(), 1, 1)
    assert ((res[True][0].cpu().double() - ref).abs() <= 2e-6 * bound + 1e-6).all()


def test_unknown_option_is_refused(ctx):
    from face_vijnana_yolov3_amd._lib import FvError
    with pytest.raises(FvError):
        ctx.set_option('no_such_switch', 1)
    with pytest.raises(FvError):
        ctx.get_option('no_such_switch')
    for key in ('overlap', 'tail_split', 'conv_waves8', 'conv1x1_persist', 'conv_bm64', 'conv_small', 'conv_halo', 'conv0_direct', 'wgrad_fused_taps'):
        assert ctx.get_option(key) == 1
        ctx.set_option(key, 0); assert ctx.get_option(key) == 0
        ctx.set_option(key, 1)


@pytest.mark.parametrize('B,H,cin,cout,k,s', [(1, 13, 512, 1024, 3, 1), (1, 26, 256, 512, 3, 1), (1, 52, 128, 256, 3, 1), (1, 26, 512, 1024, 3, 2),
                                            (3, 13, 1024, 512, 1, 1), (1, 38, 256, 512, 3, 1)])
def test_64_row_tiles_equal_128_row_tiles(ctx, B, H, cin, cout, k, s):
    """option "conv_bm64": small-M inference launches (fewer than 192 tiles of 128 x 128) whose row count pads less with 64-row
    tiles run conv_kernel<128,2,4,false,64>.  Unsplit (the operator entry point), every output element is the same k-ordered fmaf
    chain as in the 128-row tiling: bit-identical with the fused inference epilogue (affine + LeakyReLU + residual), and right."""
    from face_vijnana_yolov3_amd import ops
    x = _rand((B, H, H, cin), 311).cuda(); w = _rand((cout, k, k, cin), 312, -0.1, 0.1).cuda()
    Ho = H // s
    scale = _rand((cout,), 313, 0.5, 1.5).cuda(); shift = _rand((cout,), 314).cuda(); add = _rand((B, Ho, Ho, cout), 315).cuda()
    res = {}
    try:
        for on in (True, False):
            ctx.set_option('conv_bm64', on)
            res[on] = ops.conv2d_forward(ctx, x, w, s, scale, shift, 0.1, add)
            torch.cuda.synchronize()
    finally:
        ctx.set_option('conv_bm64', 1)
    assert torch.equal(res[True], res[False])
    ref = _ref_conv(x.cpu().double(), w.cpu().double(), k, s); bound = _ref_conv(x.cpu().double().abs(), w.cpu().double().abs(), k, s)
    pre = ref * scale.cpu().double() + shift.cpu().double()
    want = torch.where(pre > 0, pre, 0.1 * pre) + add.cpu().double()
    tol = 2e-6 * bound * scale.cpu().double().abs() + 1e-5
    # (elements within rounding of the LeakyReLU kink may take the other slope in float32: bounded by the same tolerance x 1)
    assert ((res[True].cpu().double() - want).abs() <= tol + 0.9 * (pre.abs() <= tol).double() * pre.abs()).all()


SMALL_CASES = [
    # B, H, cin, cout, k, s
    (1, 52, 256, 128, 1, 1), (1, 26, 512, 256, 1, 1), (1, 13, 1024, 512, 1, 1), (1, 26, 768, 256, 1, 1), (1, 52, 384, 128, 1, 1),   # 1x1 layers
    (1, 104, 128, 64, 1, 1),                                                            # 64 x 64 tiles, one K step per group
    (1, 52, 128, 256, 3, 2), (1, 26, 256, 512, 3, 2),                                   # the stride-2 3x3 layers into 26x26 and 13x13
    (1, 30, 128, 256, 3, 2), (2, 13, 128, 128, 3, 1), (1, 9, 128, 512, 3, 1),           # ragged tiles (225, 338, 81 rows), stride-1 3x3 with a short K
    (3, 7, 512, 1024, 1, 1),
]


@pytest.mark.parametrize('B,H,cin,cout,k,s', SMALL_CASES)
def test_small_m_conv_with_the_k_split_inside_the_workgroup(ctx, B, H, cin, cout, k, s):
    """option "conv_small" (conv_small_mfma.hip): small-M inference launches, the K steps divided over the four wave pairs / eight waves
    of a workgroup and added in group order.  Right against float64 (the operator bound), bit-reproducible, within fp32 rounding of
    the tile kernels (another summation order), for 1x1 and 3x3, stride 1 and 2, ragged M, every tile configuration."""
    from face_vijnana_yolov3_amd import ops
    Ho = H // s
    x = _rand((B, H, H, cin), 511).cuda(); w = _rand((cout, k, k, cin), 512, -0.2, 0.2).cuda()
    scale = _rand((cout,), 513, 0.5, 1.5).cuda(); shift = _rand((cout,), 514).cuda(); add = _rand((B, Ho, Ho, cout), 515).cuda()
    res = {}
    try:
        for on in (1, 0):
            ctx.set_option('conv_small', on)
            y = ops.conv2d_forward(ctx, x, w, s, scale, shift, 0.1, add)
            y2 = ops.conv2d_forward(ctx, x, w, s, scale, shift, 0.1, add)
            plain = ops.conv2d_forward(ctx, x, w, s)
            torch.cuda.synchronize()
            assert torch.equal(y, y2)
            res[on] = (y, plain)
    finally:
        ctx.set_option('conv_small', 1)
    ref = _ref_conv(x.cpu().double(), w.cpu().double(), k, s); bound = _ref_conv(x.cpu().double().abs(), w.cpu().double().abs(), k, s)
    for on in (1, 0):
        assert ((res[on][1].cpu().double() - ref).abs() <= 2e-6 * bound + 1e-6).all(), on
    assert not torch.equal(res[1][1], res[0][1])            # the new kernel really ran (another summation order)
    pre = ref * scale.cpu().double() + shift.cpu().double()
    want = torch.where(pre > 0, pre, 0.1 * pre) + add.cpu().double()
    tol = 2e-6 * bound * scale.cpu().double().abs() + 1e-5
    assert ((res[1][0].cpu().double() - want).abs() <= tol + 0.9 * (pre.abs() <= tol).double() * pre.abs()).all()
