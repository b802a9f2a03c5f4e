"""Per-launch-shape table of the batch-1 forward (fv_profile_enable(ctx, 2)): every conv launch with its tile height / K split and every
split-K finish with its slab count -- the data behind the K-split plan of the small-M path.  Usage: bs1_shapes.py [B] [S]"""
import os, sys


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from face_vijnana_yolov3_amd.engine import Engine
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 416
    eng = Engine(0); eng.init_synthetic(7)
    x = torch.rand((B, S, S, 3), device='cuda')
    for _ in range(5):
        eng.predict_device(x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        eng.predict_device(x)
    e1.record(); torch.cuda.synchronize()
    print('# %s  batch %d, %dx%d: forward %.4f ms per call (un-instrumented)' % (os.environ.get('TAG', ''), B, S, S, e0.elapsed_time(e1) / 40))
    eng.ctx.profile(True, shapes=True)
    n = 10
    for _ in range(n):
        eng.predict_device(x)
    prof = eng.ctx.profile_collect(); eng.ctx.profile(False)
    tot = 0.0
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms']):
        us = v['ms'] / n * 1e3; tot += us
        print('%-62s launches %3d  us/forward %7.1f  us/launch %6.1f' % (k, v['launches'] // n, us, v['ms'] / v['launches'] * 1e3))
    print('sum of kernel time %.1f us' % tot)


if __name__ == '__main__':
    main()
