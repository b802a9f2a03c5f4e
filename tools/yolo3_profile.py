"""Dev tool: kernel breakdown of one three-scale training step (fv_yolov3_train_step) at 416x416, batch 16."""
import os, sys

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from face_vijnana_yolov3_amd.yolov3 import Yolov3
    B, S = (int(sys.argv[2]) if len(sys.argv) > 2 else 16), 416
    m = Yolov3(0, out_channels=255); m.init_synthetic(3)
    g = torch.Generator().manual_seed(4)
    x = torch.rand((B, S, S, 3), generator=g).cuda()
    tg = [torch.rand((B, S // d, S // d, 255), generator=g).cuda() for d in (32, 16, 8)]
    for _ in range(2): m.train_on_batch(x, tg, 1e-4, 0.9, 0.999)
    torch.cuda.synchronize()
    m.ctx.set_overlap(False)
    m.ctx.profile(True, shapes=len(sys.argv) > 1)
    for _ in range(2): m.train_on_batch(x, tg, 1e-4, 0.9, 0.999)
    prof = m.ctx.profile_collect(); m.ctx.profile(False)
    tot = 0
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms']):
        ms = v['ms'] / 2; tot += ms
        print('%-34s launches %4d  ms/step %7.3f  TF %6.1f  GB/s %7.1f' % (k, v['launches'] // 2, ms, v['flops'] / (v['ms'] * 1e-3) / 1e12 if v['flops'] else 0, v['bytes'] / (v['ms'] * 1e-3) / 1e9))
    print('sum', tot)


if __name__ == '__main__':
    main()
