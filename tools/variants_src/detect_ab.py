"""detect path at batch 1 (bench.detect_bench) with the per-layer forward, the fused finish + 1x1 launch and the one-launch forward."""
import os, sys


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    from face_vijnana_yolov3_amd.engine import Engine
    eng = Engine(0); eng.init_synthetic(7)
    x = torch.rand((40, 416, 416, 3)).cuda()
    for name, persist, fuse in (('per-layer', 0, 0), ('fused finish+1x1', 0, 1), ('one launch (plan 1)', 1, 0), ('one launch (plan 2, bit-identical)', 2, 0),
                                ('per-layer', 0, 0), ('one launch (plan 1)', 1, 0)):
        eng.ctx.set_infer_persist(persist); eng.ctx.set_fuse_finish1x1(fuse)
        d = bench.detect_bench(eng, x)
        print('%-36s batch1_device %.4f  batch1_end_to_end %.4f  batch40_device %.4f' % (name, d['batch1_device'], d['batch1_end_to_end'], d['batch40_device']), flush=True)


if __name__ == '__main__':
    main()
