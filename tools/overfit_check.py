#!/usr/bin/env python3
"""Dev check: train on ONE fixed synthetic batch and watch the MSE fall (end-to-end sanity of
forward, backward and the Keras-formula Adam beyond the oracle parity tests)."""
import os, sys, time

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.engine import Engine

    B, S, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 416, int(sys.argv[2]) if len(sys.argv) > 2 else 300
    eng = Engine(0); eng.init_synthetic(seed=7)
    g = torch.Generator().manual_seed(3)
    x = torch.rand((B, S, S, 3), generator=g).cuda()
    y = torch.from_numpy(data.synth_gt_batch(B, S, seed=5)).cuda()
    t0 = time.time()
    for it in range(steps):
        loss = eng.train_on_batch(x, y, 1e-4, 0.99, 0.99)
        if it % 25 == 0 or it == steps - 1:
            print('step %4d loss %.6f' % (it, loss.item()), flush=True)
    torch.cuda.synchronize()
    print('%.1f s; finite params: %s' % (time.time() - t0, bool(torch.isfinite(eng.params).all().item())))


if __name__ == '__main__':
    main()
