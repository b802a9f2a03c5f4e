#!/usr/bin/env python3
"""Dev tool: is the batch-1 detect path host-launch-bound?  Eager vs hipGraph replay of
fv_forward_infer + fv_decode_nms (captured through torch.cuda.graph on the context's stream)."""
import os, sys, time

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from face_vijnana_yolov3_amd.engine import Engine
    from face_vijnana_yolov3_amd.postproc import decode_nms

    eng = Engine(0); eng.init_synthetic(7)
    x = torch.rand((1, 416, 416, 3), device='cuda')

    def run():
        y = eng.predict_device(x)
        return y, decode_nms(eng.ctx, y, 416, 0.5, 0.5, 60)

    def timed(fn, n=50):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    print('eager ms/img', timed(run))
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        eng.ctx.set_stream(side.cuda_stream)
        run(); side.synchronize()
        with torch.cuda.graph(g, stream=side):
            out = run()
    eng.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    print('graph ms/img', timed(g.replay))
    y_ref, r_ref = run(); torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print('graph == eager:', torch.equal(out[0], y_ref), torch.equal(out[1]['boxes'], r_ref['boxes']))


if __name__ == '__main__':
    main()
