"""What does the data-parallel machinery cost on ONE GPU?  ms/step of (a) the plain step, (b) the bucket path (fv_bucket_fn
callbacks -> event -> comm stream) without a process group, (c) the same with a world-size-1 nccl group (real RCCL calls),
at several bucket sizes.  Diagnostic for the N > 1 scaling efficiency (bench.py multi_gpu)."""
import os, sys, time

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


    def run(tr, n=20, label=''):
        for _ in range(3):
            tr.train_on_batch(x, y, **HPS)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            tr.train_on_batch(x, y, **HPS)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        # host time of one step's enqueue (no sync inside): how far ahead of the GPU the host runs
        t0 = time.perf_counter(); tr.train_on_batch(x, y, **HPS); host = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        print('%-44s %.2f ms/step   (host enqueue of one step %.2f ms)' % (label, ms, host), flush=True)


    run(DataParallelTrainer(eng, world_size=1, rank=0), label='plain')
    for mib in (32, 256):
        run(DataParallelTrainer(eng, world_size=1, rank=0, bucket_bytes=mib << 20, force_bucket_path=True), label='bucket path, no group, %d MiB' % mib)
    dist.init_process_group('nccl', store=dist.HashStore(), rank=0, world_size=1, device_id=eng.dev)
    for mib in (32, 256):
        run(DataParallelTrainer(eng, world_size=1, rank=0, bucket_bytes=mib << 20, force_bucket_path=True), label='bucket path, RCCL world 1, %d MiB' % mib)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
