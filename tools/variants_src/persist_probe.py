"""One-launch small-M forward (infer_persist.hip) against the per-layer path: equality, time per image, per-phase trace."""
import os
import sys
import time


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from face_vijnana_yolov3_amd.engine import Engine
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 416
    grids = [int(g) for g in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0]
    eng = Engine(0); eng.init_synthetic(7)
    x = torch.rand((1, S, S, 3), generator=torch.Generator().manual_seed(1)).cuda()

    def timed(n=50):
        eng.predict_device(x); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            eng.predict_device(x)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    eng.ctx.set_infer_persist(0)
    eng.ctx.set_fuse_finish1x1(False)
    y0 = eng.predict_device(x).clone(); torch.cuda.synchronize()
    print('per-layer path        %.4f ms/img' % timed(), flush=True)
    eng.ctx.set_fuse_finish1x1(True)
    yf = eng.predict_device(x).clone(); torch.cuda.synchronize()
    print('per-layer path, finish + 1x1 fused: bit-identical %s   %.4f ms/img' % (torch.equal(yf, y0), timed()), flush=True)
    if len(sys.argv) > 3:
        return
    for grid in grids:
        for mode in (2, 1):
            eng.ctx.set_infer_persist(mode, grid)
            y = eng.predict_device(x).clone()
            eng.ctx.infer_persist_status()
            y2 = eng.predict_device(x).clone(); torch.cuda.synchronize()
            d = (y - y0).abs().max().item() / y0.abs().max().item()
            print('persist mode %d grid %3d: max rel diff to per-layer %.3e  bit-identical %s  repeatable %s' % (
                mode, grid, d, torch.equal(y, y0), torch.equal(y, y2)), flush=True)
            print('                        %.4f ms/img' % timed(), flush=True)
            eng.ctx.infer_persist_trace(True)
            eng.predict_device(x); torch.cuda.synchronize()
            t = eng.ctx.infer_persist_trace(False, read=True)
            nph = (len(t) - 1) // 3
            tot_a = sum(t[3 * i + 1] - t[3 * i] for i in range(nph)); tot_b = sum(t[3 * i + 2] - t[3 * i + 1] for i in range(nph))
            tot_c = sum(t[3 * i + 3] - t[3 * i + 2] for i in range(nph))
            print('   trace (workgroup 0): launch %.1f us = tile loops %.1f + slab reductions %.1f + barriers %.1f' % (t[-1], tot_a, tot_b, tot_c))
            L = eng.layers
            for i in range(nph):
                d = L[9 + i]
                print('     L%02d %dx%d s%d %4d->%4d @%3d: tiles+slices %6.1f  reduce %5.1f  barrier %5.1f' % (
                    9 + i, d['ksize'], d['ksize'], d['stride'], d['cin'], d['cout'], S // d['out_div'],
                    t[3 * i + 1] - t[3 * i], t[3 * i + 2] - t[3 * i + 1], t[3 * i + 3] - t[3 * i + 2]))
    eng.ctx.set_infer_persist(1, 0)


if __name__ == '__main__':
    main()
