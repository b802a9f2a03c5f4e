"""Why does a base training step run at 60 instead of 53 ms when its engine is created late in a process?  (tools/stream_env_probe.py)"""
import os, sys, time

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.engine import Engine
    from face_vijnana_yolov3_amd.yolov3 import Yolov3
    HPS = dict(lr=1e-4, beta_1=0.99, beta_2=0.99, decay=0.0)
    x40 = torch.rand((40, 416, 416, 3)).cuda(); y40 = torch.from_numpy(data.synth_gt_batch(40, 416, seed=1)).cuda()
    def base(eng, label, n=8):
        for _ in range(3):
            eng.train_on_batch(x40, y40, **HPS)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            eng.train_on_batch(x40, y40, **HPS)
        torch.cuda.synchronize()
        print('%-64s %.2f ms/step' % (label, (time.perf_counter() - t0) / n * 1e3), flush=True)
    e1 = Engine(0); e1.init_synthetic(7)
    base(e1, 'A  engine 1, fresh process')
    m = Yolov3(0, out_channels=255); m.init_synthetic(3)
    g = torch.Generator().manual_seed(4)
    x16 = torch.rand((16, 416, 416, 3), generator=g).cuda()
    tg = [torch.rand((16, 416 // d, 416 // d, 255), generator=g).cuda() for d in (32, 16, 8)]
    for _ in range(3):
        m.train_on_batch(x16, tg, 1e-4, 0.9, 0.99)
    torch.cuda.synchronize()
    base(e1, 'B  engine 1 again, after a three-scale model trained')
    e2 = Engine(0); e2.init_synthetic(7)
    base(e2, 'C  engine 2, created late (own context, own workspace)')
    e2.ctx.set_overlap(False)
    base(e2, 'D  engine 2, weight-gradients on the main stream')
    e2.ctx.set_overlap(True)
    e1.ctx.set_overlap(False)
    base(e1, 'E  engine 1, weight-gradients on the main stream')
    e1.ctx.set_overlap(True)
    # engine 2's tensors inside engine 1's context: is it the memory or the context?
    e2.ctx, keep = e1.ctx, e2.ctx
    base(e2, 'F  engine 2 tensors driven through engine 1 context')
    e2.ctx = keep
    print(torch.cuda.memory_allocated() / 2**30, 'GiB allocated,', torch.cuda.memory_reserved() / 2**30, 'GiB reserved')


if __name__ == '__main__':
    main()
