"""N>1 host logic on CPU, world_size 2 over gloo: the tower slicing with its weights, the results-directory
reset that must not race with other ranks' writes, and the merge of per-rank csv parts into the single
6-column solution file the reference's evaluate()/test() leave behind (face_detection.py:716-738, 857-875)."""
import os

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from face_vijnana_yolov3_amd.parallel import merge_rank_files, part_path, reset_dir_before_shards, shard_files, slice_batch


def test_slice_batch_is_multi_gpu_models_split():
    assert slice_batch(40, 4, 0) == (0, 10, 0.25) and slice_batch(40, 4, 3) == (30, 40, 0.25)
    # remainder to the last tower, weights n_r / n
    assert slice_batch(7, 2, 0) == (0, 3, 3 / 7) and slice_batch(7, 2, 1) == (3, 7, 4 / 7)
    assert abs(sum(slice_batch(37, 8, r)[2] for r in range(8)) - 1.0) < 1e-12
    # fewer images than ranks: skipped on EVERY rank (an empty tower would stay out of the collectives)
    assert all(slice_batch(3, 8, r) is None for r in range(8))
    assert part_path('a.csv', 1, 0) == 'a.csv' and part_path('a.csv', 2, 1) == 'a.csv.rank1'


def _worker(rank, world, port, tmp):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    res = os.path.join(tmp, 'results')
    if rank == 0:                                   # leftovers of an earlier run
        os.makedirs(res, exist_ok=True)
        open(os.path.join(res, 'stale_detected.jpg'), 'w').write('x')
    dist.barrier()
    reset_dir_before_shards(res, rank)
    files = ['img_%03d.jpg' % i for i in range(7)]
    mine = shard_files(files, world, rank)
    for f in mine:                                  # every rank writes right away: nothing may be deleted under it
        open(os.path.join(res, f[:-4] + '_detected.jpg'), 'w').write(f)
    out = os.path.join(tmp, 'solution.csv')
    with open(part_path(out, world, rank), 'w') as f:
        for name in mine:
            f.write('%s,1,2,3,4,0.5\n' % name)
    merge_rank_files(out, world, rank)
    rat = os.path.join(tmp, 'ratios.csv')
    with open(part_path(rat, world, rank), 'w') as f:
        f.write('ratio\n')
        for name in mine:
            f.write('%d.5\n' % int(name[4:7]))
    merge_rank_files(rat, world, rank, header_lines=1)
    dist.barrier()
    dist.destroy_process_group()


def test_world2_reset_and_merge(tmp_path):
    port = 29000 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = sorted(os.listdir(tmp_path / 'results'))
    assert res == ['img_%03d_detected.jpg' % i for i in range(7)]          # stale file gone, nobody's output lost
    rows = open(tmp_path / 'solution.csv').read().splitlines()
    assert [r.split(',')[0] for r in rows] == ['img_%03d.jpg' % i for i in range(7)]   # single-process row order
    assert all(len(r.split(',')) == 6 for r in rows)
    assert open(tmp_path / 'ratios.csv').read().splitlines() == ['ratio'] + ['%d.5' % i for i in range(7)]
    assert not [f for f in os.listdir(tmp_path) if '.rank' in f]
