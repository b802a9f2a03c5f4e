"""Per-launch-shape breakdown of the base training step in the serial schedule: fv_profile_enable(ctx, 2).  Usage: base_profile.py [batch=40] [image_size=416]"""
import os, sys


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.engine import Engine
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 416
    eng = Engine(0); eng.init_synthetic(7)
    x = torch.rand((B, S, S, 3)).cuda(); y = torch.from_numpy(data.synth_gt_batch(B, S, seed=1)).cuda()
    for _ in range(3):
        eng.train_on_batch(x, y, **bench.HPS)
    eng.ctx.set_overlap(False)
    eng.train_on_batch(x, y, **bench.HPS); torch.cuda.synchronize()
    eng.ctx.profile(True, shapes=True)
    n = 3
    for _ in range(n):
        eng.train_on_batch(x, y, **bench.HPS)
    prof = eng.ctx.profile_collect(); eng.ctx.profile(False)
    tot = 0.0
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms']):
        ms = v['ms'] / n; tot += ms
        print('%-58s launches %3d  ms/step %7.3f  us/launch %7.1f  TF %6.1f  GB/s %7.1f' % (
            k, v['launches'] // n, ms, v['ms'] / v['launches'] * 1e3, v['flops'] / (v['ms'] * 1e-3) / 1e12 if v['flops'] else 0, v['bytes'] / (v['ms'] * 1e-3) / 1e9))
    print('sum %.3f ms' % tot)


if __name__ == '__main__':
    main()
