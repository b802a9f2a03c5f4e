"""Which process state slows a training step that is fast in a fresh process?  (bench.py measured the three-scale step at 41 ms
after its other measurements against 31.6 ms standalone.)  Step time and host enqueue time of the three-scale step after each of:
torch's stream pool coming to life, an RCCL group created and destroyed, further contexts, a FaceDetector.test() loop."""
import os, sys, time, tempfile

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch, torch.distributed as dist
    from face_vijnana_yolov3_amd.yolov3 import Yolov3
    from face_vijnana_yolov3_amd.engine import Engine
    m = Yolov3(0, out_channels=255); m.init_synthetic(3)
    g = torch.Generator().manual_seed(4)
    x = torch.rand((16, 416, 416, 3), generator=g).cuda()
    tg = [torch.rand((16, 416 // d, 416 // d, 255), generator=g).cuda() for d in (32, 16, 8)]
    def measure(label):
        for _ in range(2):
            m.train_on_batch(x, tg, 1e-4, 0.9, 0.99)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            m.train_on_batch(x, tg, 1e-4, 0.9, 0.99)
        host = (time.perf_counter() - t0) / 5 * 1e3
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print('%-50s step %.2f ms   host enqueue %.2f ms' % (label, ms, host), flush=True)
    measure('fresh process')
    s = torch.cuda.Stream(); s.synchronize()
    measure('after torch.cuda.Stream() (pool of 32 streams)')
    dist.init_process_group('nccl', store=dist.HashStore(), rank=0, world_size=1, device_id=torch.device('cuda', 0))
    t = torch.ones(1024, device='cuda'); dist.all_reduce(t); torch.cuda.synchronize()
    measure('with a world-size-1 nccl group alive')
    dist.destroy_process_group()
    measure('after destroy_process_group')
    engs = [Engine(0) for _ in range(3)]
    for e in engs:
        e.init_synthetic(1); e.predict_device(x[:1])
    measure('after three more contexts')
    import bench
    bench.test_loop_bench(0, 416, n_img=16)
    measure('after a FaceDetector.test() loop')
    from face_vijnana_yolov3_amd import data
    eng = engs[0]
    x40 = torch.rand((40, 416, 416, 3)).cuda(); y40 = torch.from_numpy(data.synth_gt_batch(40, 416, seed=1)).cuda()
    for _ in range(3):
        eng.train_on_batch(x40, y40, **bench.HPS)
    torch.cuda.synchronize()
    measure('after base training steps (13 GB workspace alive)')
    print(bench.rccl_world1_rehearsal(eng, x40, y40))
    measure('after bench.rccl_world1_rehearsal')


if __name__ == '__main__':
    main()
