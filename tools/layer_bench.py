#!/usr/bin/env python3
"""Per-layer-shape timing of the MFMA kernels at the BASELINE batch (dev tool, not a test).

    python tools/layer_bench.py [--batch 40] [--size 416] [--reps 5]
Prints TFLOP/s of forward / data-gradient / weight-gradient for every distinct conv shape of
Darknet-53 (HIP events on the launch stream, through the C ABI)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from face_vijnana_yolov3_amd import ops  # noqa: E402
from face_vijnana_yolov3_amd._lib import Context  # noqa: E402
from face_vijnana_yolov3_amd.engine import layer_table  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=40)
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--only', default='')
    ap.add_argument('--scratch-mib', type=int, default=0, help='lend conv scratch (enables the tail split at op level)')
    ap.add_argument('--waves8', type=int, default=1, help='0: the 4-wave forms of the 128- and 64-wide conv tiles')
    a = ap.parse_args()
    ctx = Context(0)
    ctx.set_conv_waves8(bool(a.waves8))
    if a.scratch_mib:
        ctx.set_conv_scratch(torch.empty(a.scratch_mib << 20, dtype=torch.uint8, device='cuda'))
    seen = {}
    for d in layer_table():
        key = (d['ksize'], d['stride'], d['cin'], d['cout'], a.size // d['in_div'])
        seen[key] = seen.get(key, 0) + 1
    print('%-28s %5s | %8s %7s | %8s %7s | %8s %7s' % ('k s cin cout H', 'count', 'fwd ms', 'TF', 'dgrad ms', 'TF', 'wgrad ms', 'TF'))
    tot = [0.0, 0.0, 0.0]
    for (k, s, cin, cout, H), cnt in seen.items():
        if a.only and ('%d_%d_%d_%d' % (k, s, cin, cout)) not in a.only.split(','):
            continue
        B = a.batch
        x = torch.rand((B, H, H, cin), device='cuda')
        w = torch.rand((cout, k, k, cin), device='cuda') - 0.5
        Ho = H // s
        cp = max(32, cout)
        dy = torch.rand((B, Ho, Ho, cp), device='cuda')
        flops = 2.0 * B * Ho * Ho * cout * k * k * cin
        wd = ops.pack_first_layer(ctx, w) if cin % 32 else w
        out = torch.empty((B, Ho, Ho, cout), device='cuda')
        from face_vijnana_yolov3_amd._lib import lib, ptr, c_void_p
        NULL = c_void_p(None)
        L = lib()
        t_f = timeit(lambda: L.fv_conv2d_forward(ctx.handle, ptr(x), ptr(wd), B, H, H, cin, cout, k, s, NULL, NULL, -1.0, NULL, ptr(out), NULL, NULL), a.reps)
        if cin % 32 == 0:
            wt = ops.transpose_weights(ctx, w, cp)
            dx = torch.empty_like(x)
            t_d = timeit(lambda: L.fv_conv2d_dgrad(ctx.handle, ptr(dy), ptr(wt), B, H, H, cin, cp, k, s, NULL, ptr(dx)), a.reps)
        else:
            t_d = float('nan')
        dw = torch.zeros_like(w)
        t_w = timeit(lambda: L.fv_conv2d_wgrad(ctx.handle, ptr(x), ptr(dy), B, H, H, cin, cout, cp, k, s, ptr(dw)), a.reps)
        tf = lambda t: flops / (t * 1e-3) / 1e12
        print('%d %d %4d %4d %3d %14s %5d | %8.3f %7.1f | %8.3f %7.1f | %8.3f %7.1f' % (k, s, cin, cout, H, '', cnt, t_f, tf(t_f), t_d, tf(t_d), t_w, tf(t_w)))
        tot[0] += t_f * cnt; tot[1] += (0 if t_d != t_d else t_d * cnt); tot[2] += t_w * cnt
    print('sum over network (ms): fwd %.2f  dgrad %.2f  wgrad %.2f  total %.2f' % (tot[0], tot[1], tot[2], sum(tot)))


if __name__ == '__main__':
    main()
