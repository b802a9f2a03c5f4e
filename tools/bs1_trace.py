"""Dev tool: per-launch durations of one batch-1 forward (run under rocprofv3 --kernel-trace)."""
import os, sys

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from face_vijnana_yolov3_amd.engine import Engine
    eng = Engine(0); eng.init_synthetic(seed=7)
    x = torch.rand((1, 416, 416, 3), device='cuda')
    for _ in range(3): eng.predict_device(x)
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
