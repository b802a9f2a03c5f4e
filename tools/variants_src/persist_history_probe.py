"""Why does the one-launch forward take 2.1 ms inside bench.py and 1.27 ms in a process that only runs inference?"""
import os, sys


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.engine import Engine
    eng = Engine(0); eng.init_synthetic(7)
    x1 = torch.rand((1, 416, 416, 3)).cuda()
    x40 = torch.rand((40, 416, 416, 3)).cuda(); y40 = torch.from_numpy(data.synth_gt_batch(40, 416, seed=1)).cuda()

    def timed(n=30):
        eng.predict_device(x1); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            eng.predict_device(x1)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    def both(label):
        eng.ctx.set_infer_persist(0); a = timed()
        eng.ctx.set_infer_persist(1); b = timed(); eng.ctx.infer_persist_status()
        eng.ctx.set_infer_persist(0)
        print('%-52s per-layer %.4f   one launch %.4f ms/img' % (label, a, b), flush=True)

    both('fresh')
    eng.predict_device(x40); torch.cuda.synchronize()
    both('after a batch-40 forward (second inference workspace)')
    for _ in range(3):
        eng.train_on_batch(x40, y40, **bench.HPS)
    torch.cuda.synchronize()
    both('after 3 training steps (side stream used)')
    eng.ctx.set_overlap(False)
    eng.train_on_batch(x40, y40, **bench.HPS); torch.cuda.synchronize()
    eng.ctx.set_overlap(True)
    both('after a serial training step')
    eng.ctx.profile(True); eng.train_on_batch(x40, y40, **bench.HPS); eng.ctx.profile_collect(); eng.ctx.profile(False)
    both('after an instrumented step (event pool)')
    print(bench.rccl_world1_rehearsal(eng, x40, y40, steps=3))
    both('after the world-1 collective rehearsal')


if __name__ == '__main__':
    main()
