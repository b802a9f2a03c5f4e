"""The three-scale training step alone (bench.three_scale_bench), with the per-kernel profile."""
import os, sys

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    print(bench.three_scale_bench(0, 416, steps=10))
    from face_vijnana_yolov3_amd.yolov3 import Yolov3
    m = Yolov3(0, out_channels=255); m.init_synthetic(3)
    g = torch.Generator().manual_seed(4)
    x = torch.rand((16, 416, 416, 3), generator=g).cuda()
    tg = [torch.rand((16, 416 // d, 416 // d, 255), generator=g).cuda() for d in (32, 16, 8)]
    for ov in (True, False):
        m.ctx.set_overlap(ov)
        for _ in range(2):
            m.train_on_batch(x, tg, 1e-4, 0.9, 0.99)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            m.train_on_batch(x, tg, 1e-4, 0.9, 0.99)
        e1.record(); torch.cuda.synchronize()
        print('overlap', ov, 'ms/step', e0.elapsed_time(e1) / 5)
    m.ctx.set_overlap(False)
    m.ctx.profile(True)
    for _ in range(2):
        m.train_on_batch(x, tg, 1e-4, 0.9, 0.99)
    p = m.ctx.profile_collect(); m.ctx.profile(False)
    for k, v in sorted(p.items(), key=lambda kv: -kv[1]['ms'])[:14]:
        print('%-34s launches/step %5.1f  ms/step %.3f' % (k, v['launches'] / 2, v['ms'] / 2))


if __name__ == '__main__':
    main()
