"""Device-only rate of the test()/evaluate() batch (fv_forward_infer + fv_decode_nms, HIP events) against hps.eval_batch_size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_vijnana_yolov3_amd.engine import Engine
from face_vijnana_yolov3_amd.postproc import decode_nms

def main():
    for S in (416, 608):
        eng = Engine(0); eng.init_synthetic(7)
        for B in (8, 16, 24, 32, 40, 48, 56, 64, 80, 96):
            if S == 608 and B > 48: break
            x = torch.rand((B, S, S, 3), device='cuda')
            def once():
                y = eng.predict_device(x)
                return decode_nms(eng.ctx, y, S, 0.5, 0.5, 60)
            for _ in range(3): once()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): once()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 8
            print('S %d  batch %3d: %7.3f ms  %.4f ms/img  %7.1f img/s' % (S, B, ms, ms / B, B * 1e3 / ms), flush=True)

main()
