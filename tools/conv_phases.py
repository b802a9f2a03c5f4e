#!/usr/bin/env python3
"""Where a conv launch's time goes, per workgroup (dev tool; needs the stamps build):

    tools/build_variant.sh stamps -DFV_CONV_STAMPS
    FV_LIB_PATH=tools/_variants/libfv_stamps.so python tools/conv_phases.py

Every workgroup of conv_kernel stamps wall_clock64 (100 MHz) at entry, after its first operand tile is staged, after its K loop
and after its last store, plus its hardware id.  For each 1x1 shape of the base step (training forward with statistics slots;
data-gradient with the fused BN-backward reduction and the residual addend) this prints the launch's span, the per-phase
medians, the start skew and what a CU's two slots did over time."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from face_vijnana_yolov3_amd import ops  # noqa: E402
from face_vijnana_yolov3_amd._lib import Context, lib  # noqa: E402


def stamps(ctx, nwg, fn='fv_debug_conv_stamps'):
    buf = np.zeros((nwg, 5), np.uint64)
    f = getattr(lib(), fn)
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    ctx.check(f(ctx.handle, buf.ctypes.data_as(ctypes.c_void_p), nwg), fn)
    return buf


def report(name, st, flops):
    t = st[:, :4].astype(np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) * 0.01
    hw = st[:, 4]
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
    hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
    cu = (hwid >> 8) & 0xf
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    cuid = xcc * 256 + se * 32 + sh * 16 + cu
    span = us[:, 3].max()
    pro, kl, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
    print('%s: %d workgroups on %d CUs, span %.1f us = %.1f TF' % (name, len(st), len(set(cuid.tolist())), span, flops / span / 1e6))
    q = lambda v: '%.1f / %.1f / %.1f' % tuple(np.percentile(v, [10, 50, 90]))
    print('   start (p10/p50/p90) %s   prologue %s   K loop %s   epilogue %s   end %s' % (q(us[:, 0]), q(pro), q(kl), q(epi), q(us[:, 3])))
    # rounds: workgroups ordered by start time; how many are in the K loop at a time (sampled)
    grid = np.linspace(0, span, 41)
    ink = [(int(((us[:, 1] <= g) & (us[:, 2] > g)).sum()), int(((us[:, 0] <= g) & (us[:, 3] > g)).sum())) for g in grid]
    print('   resident / in K loop over time:', ' '.join('%d/%d' % (r, k) for k, r in ink[::2]))
    # one CU's slots
    c0 = np.bincount(cuid).argmax()
    rows = sorted((us[i, 0], us[i, 1], us[i, 2], us[i, 3]) for i in np.nonzero(cuid == c0)[0])
    print('   CU %d: ' % c0 + '  '.join('[%.1f %.1f %.1f %.1f]' % r for r in rows[:8]))


def main():
    ctx = Context(0)
    ctx.set_conv_scratch(torch.empty(64 << 20, dtype=torch.uint8, device='cuda'))
    ctx.set_option('conv1x1_persist', 0)        # the stamps are per workgroup = per tile in the one-tile kernel only
    B = 40
    g = torch.Generator(device='cuda').manual_seed(0)
    for (H, cin, cout) in ((52, 256, 128), (26, 512, 256), (13, 1024, 512)):
        M = B * H * H
        x = torch.rand((B, H, H, cin), device='cuda', generator=g)
        w = torch.rand((cout, 1, 1, cin), device='cuda', generator=g) - 0.5
        # training forward of the 1x1 layer (cin -> cout) with statistics slots
        slots = ops.stat_slots(cout, 'cuda')
        for _ in range(3):
            ops.conv2d_forward_slots(ctx, x, w, 1, slots)
        torch.cuda.synchronize()
        nwg = ((M + 127) // 128) * ((cout + 127) // 128) * 3
        report('fwd 1x1 M%d N%d K%d' % (M, cout, cin), stamps(ctx, min(nwg, 32768))[:((M + 127) // 128) * ((cout + 127) // 128)], 2.0 * M * cin * cout)
        # its data-gradient: dy (cout ch) -> dx (cin ch), + addend, + fused BN-backward reduction of the producer (cin ch)
        dy = torch.rand((B, H, H, cout), device='cuda', generator=g)
        z = torch.rand((B, H, H, cin), device='cuda', generator=g)
        add = torch.rand((B, H, H, cin), device='cuda', generator=g)
        vec = lambda: torch.rand(cin, device='cuda', generator=g) + 0.5
        sc, sh, mu, isd = vec(), vec(), vec(), vec()
        bs = ops.stat_slots(cin, 'cuda')
        for _ in range(3):
            ops.conv2d_dgrad_bnred(ctx, dy, w, (H, H), 1, z, sc, sh, mu, isd, bs, addend=add)
        torch.cuda.synchronize()
        nt = ((M + 127) // 128) * ((cin + 127) // 128)
        report('dgrad 1x1 r M%d N%d K%d' % (M, cin, cout), stamps(ctx, min(nt * 3, 32768))[:nt], 2.0 * M * cin * cout)
    # weight-gradients of the 1x1 layers (x: cin channels, dy: cout channels)
    for (H, cin, cout) in ((52, 256, 128), (26, 512, 256), (13, 1024, 512), (52, 128, 256)):
        M = B * H * H
        x = torch.rand((B, H, H, cin), device='cuda', generator=g)
        dy = torch.rand((B, H, H, cout), device='cuda', generator=g)
        k = 1 if (cin, cout) != (128, 256) else 3
        for _ in range(3):
            ops.conv2d_wgrad(ctx, x, dy, cout, k)
        torch.cuda.synchronize()
        time.sleep(0.02)                                 # 2e6 ticks: the measured launch's stamps are the only ones of the last 2 ms
        ops.conv2d_wgrad(ctx, x, dy, cout, k)
        torch.cuda.synchronize()
        st = stamps(ctx, 8192, 'fv_debug_wgrad_stamps')
        st = st[st[:, 3] > st[:, 0]]
        live = st[(st[:, 0] + np.uint64(200000) >= st[:, 0].max())]
        report('wgrad %dx%d M%d N%d C%d' % (k, k, M, cout, cin), live, 2.0 * M * cin * cout * k * k)
    # a 3x3 layer for scale
    H, cin, cout = 52, 128, 256
    M = B * H * H
    x = torch.rand((B, H, H, cin), device='cuda', generator=g)
    w = torch.rand((cout, 3, 3, cin), device='cuda', generator=g) - 0.5
    slots = ops.stat_slots(cout, 'cuda')
    for _ in range(2):
        ops.conv2d_forward_slots(ctx, x, w, 1, slots)
    torch.cuda.synchronize()
    nt = ((M + 127) // 128) * ((cout + 127) // 128)
    report('fwd 3x3 M%d N%d K%d (whole tiles only)' % (M, cout, 9 * cin), stamps(ctx, nt)[:nt], 2.0 * M * 9 * cin * cout)


if __name__ == '__main__':
    main()
