"""Does something that happens BEFORE a model's workspace is allocated slow its steps?  One candidate per process."""
import os, sys, time

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch, torch.distributed as dist
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.engine import Engine
    HPS = dict(lr=1e-4, beta_1=0.99, beta_2=0.99, decay=0.0)
    what = sys.argv[1]
    torch.cuda.init()
    if what == 'rccl':
        dist.init_process_group('nccl', store=dist.HashStore(), rank=0, world_size=1, device_id=torch.device('cuda', 0))
        t = torch.ones(1024, device='cuda'); dist.all_reduce(t); torch.cuda.synchronize()
    elif what == 'rccl_destroyed':
        dist.init_process_group('nccl', store=dist.HashStore(), rank=0, world_size=1, device_id=torch.device('cuda', 0))
        t = torch.ones(1024, device='cuda'); dist.all_reduce(t); torch.cuda.synchronize()
        dist.destroy_process_group()
    elif what == 'pinned':
        keep = [torch.empty(150 << 20, dtype=torch.uint8).pin_memory() for _ in range(3)]
    elif what == 'streams':
        s = torch.cuda.Stream(); s.synchronize()
    elif what == 'test_loop':
        import bench
        bench.test_loop_bench(0, 416, n_img=16)
    x40 = torch.rand((40, 416, 416, 3)).cuda(); y40 = torch.from_numpy(data.synth_gt_batch(40, 416, seed=1)).cuda()
    eng = Engine(0); eng.init_synthetic(7)
    for _ in range(3):
        eng.train_on_batch(x40, y40, **HPS)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        eng.train_on_batch(x40, y40, **HPS)
    torch.cuda.synchronize()
    print('%-16s base step %.2f ms' % (what, (time.perf_counter() - t0) / 10 * 1e3), flush=True)


if __name__ == '__main__':
    main()
