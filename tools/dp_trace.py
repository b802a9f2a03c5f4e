"""Steps of the data-parallel trainer with a world-size-1 RCCL group in one comm mode (argv[1]: pg | main | side), to be run under
`rocprofv3 --kernel-trace --output-format csv`: the kernel timeline then shows where a non-blocking collective costs its time."""
import os, sys

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch, torch.distributed as dist
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.engine import Engine
    from face_vijnana_yolov3_amd.parallel import DataParallelTrainer
    HPS = dict(lr=1e-4, beta_1=0.99, beta_2=0.99, decay=0.0)
    eng = Engine(0); eng.init_synthetic(7)
    B, S = 40, 416
    x = torch.rand((B, S, S, 3)).cuda(); y = torch.from_numpy(data.synth_gt_batch(B, S, seed=1)).cuda()
    dist.init_process_group('nccl', store=dist.HashStore(), rank=0, world_size=1, device_id=eng.dev)
    tr = DataParallelTrainer(eng, world_size=1, rank=0, force_bucket_path=True, comm_mode=sys.argv[1])
    import time
    for _ in range(4):
        tr.train_on_batch(x, y, **HPS)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        tr.train_on_batch(x, y, **HPS)
    torch.cuda.synchronize()
    print('%s: %.2f ms per step' % (sys.argv[1], (time.perf_counter() - t0) / 4 * 1e3), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
