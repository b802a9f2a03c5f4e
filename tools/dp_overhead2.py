"""Per-step times of the bucket path with / without a world-size-1 RCCL group (FV_COMM_STREAM=pg|side|main selects where the
collectives run, parallel.DataParallelTrainer)."""
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
    def steps(fn, n):
        out = []
        for _ in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 1e3)
        return out
    print('plain      ', ' '.join('%.1f' % t for t in steps(lambda: eng.train_on_batch(x, y, **HPS), 12)), flush=True)
    mode = sys.argv[1] if len(sys.argv) > 1 else 'rccl'
    if mode == 'rccl_after':        # trainer (and its comm stream) first, then the group -- DataParallelTrainer's own order when world > 1
        tr = DataParallelTrainer(eng, world_size=1, rank=0, force_bucket_path=True)
        dist.init_process_group('nccl', store=dist.HashStore(), rank=0, world_size=1, device_id=eng.dev)
        tr.collective = True
    else:
        if mode == 'rccl':
            dist.init_process_group('nccl', store=dist.HashStore(), rank=0, world_size=1, device_id=eng.dev)
        tr = DataParallelTrainer(eng, world_size=1, rank=0, force_bucket_path=True, bucket_bytes=int(os.environ.get('FV_BUCKET_MIB', '32')) << 20)
    print('bucket %-5s' % mode, ' '.join('%.1f' % t for t in steps(lambda: tr.train_on_batch(x, y, **HPS), 16)), flush=True)
    print('plain again', ' '.join('%.1f' % t for t in steps(lambda: eng.train_on_batch(x, y, **HPS), 8)), flush=True)


if __name__ == '__main__':
    main()
