"""N>1 host logic on CPU: world_size-2 gloo run of the gradient bucket reducer (the same class the
RCCL path drives from the fv_bucket_fn callback)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, tmp):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from face_vijnana_yolov3_amd.parallel import BucketReducer
    from face_vijnana_yolov3_amd.engine import layer_table
    layers = layer_table()
    n = layers[-1]['beta_off'] + 6
    n_small = 200000  # emulate the layout on a reduced vector: ranges scaled down
    scale = n_small / n
    flat = torch.full((n_small,), float(rank + 1))
    launched = []

    def launch(view):
        dist.all_reduce(view, op=dist.ReduceOp.SUM)
        view.mul_(1.0 / world)
        launched.append(view.numel())

    red = BucketReducer(flat, world, bucket_bytes=64 * 1024, launch=launch)
    # ranges arrive in reverse layer order, contiguous, exactly like fv_train_step reports them
    bounds = sorted({int(d['w_off'] * scale) for d in layers} | {n_small})
    ranges = [(bounds[i], bounds[i + 1] - bounds[i]) for i in range(len(bounds) - 1)][::-1]
    for off, cnt in ranges:
        if cnt:
            red.on_range(off, cnt)
    red.flush()
    ok = bool(torch.allclose(flat, torch.full_like(flat, (1 + world) / 2.0)))
    cover = sorted(red.launched)
    ok = ok and cover[0][0] == 0 and cover[-1][1] == n_small and all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))
    ok = ok and len(launched) < len(ranges)          # coalesced into fewer, larger collectives
    with pytest.raises(RuntimeError):
        red.reset(); red.on_range(100, 10); red.on_range(500, 10)   # not contiguous-descending
    np.save(os.path.join(tmp, 'ok%d.npy' % rank), np.array([ok, len(launched)]))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_reducer_world2_gloo(tmp_path):
    world = 2
    port = 29000 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok, nl = np.load(os.path.join(str(tmp_path), 'ok%d.npy' % r))
        assert ok == 1 and nl >= 2
