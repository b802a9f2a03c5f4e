"""Real RCCL calls on the one GPU of the test box: a `nccl` process group of world size 1 drives
DataParallelTrainer's bucket path -- fv_bucket_fn callbacks -> event -> communication stream -> dist.all_reduce (RCCL) -> wait ->
Adam -- and must reproduce the plain single-GPU step (reference behaviour replaced: keras.utils.multi_gpu_model,
face_detection.py:358-371, 612-619).  Runs in a child process so that the group does not outlive the test; also `bench.py
--gpus 1 --spawn` (the self-launcher with real devices) must print the driver's JSON line with `multi_gpu.rccl_ranks` == 1."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from face_vijnana_yolov3_amd.engine import Engine
from face_vijnana_yolov3_amd.parallel import DataParallelTrainer
eng = Engine(0)
g = torch.Generator().manual_seed(3)
B, S = 4, 96
x = torch.rand((B, S, S, 3), generator=g).cuda(); y = torch.rand((B, S // 32, S // 32, 6), generator=g).cuda()
res = []
for mode in ('plain', 'rccl'):
    eng.init_synthetic(seed=7)
    eng.iterations = 0; eng.m = eng.v = eng.grads = None
    if mode == 'rccl':
        dist.init_process_group('nccl', store=dist.HashStore(), rank=0, world_size=1, device_id=eng.dev)
    tr = DataParallelTrainer(eng, world_size=1, rank=0, bucket_bytes=8 << 20, force_bucket_path=(mode == 'rccl'))
    assert tr.collective == (mode == 'rccl')
    loss1 = float(tr.train_on_batch(x, y, 1e-4, 0.99, 0.99).item())
    torch.cuda.synchronize()
    g1 = eng.grads.clone(); m1 = eng.m.clone()
    for _ in range(2):
        loss = tr.train_on_batch(x, y, 1e-4, 0.99, 0.99)
    torch.cuda.synchronize()
    ncoll = tr.collectives_launched
    res.append((loss1, g1, m1, float(loss.item()), eng.state.clone(), ncoll))
    if mode == 'rccl':
        cover = sorted(tr.reducer.launched)
        assert cover[0][0] == 0 and cover[-1][1] == eng.n_params and len(cover) >= 5
        assert all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))
        assert ncoll == 3 * (len(cover) + 1), (ncoll, len(cover))    # every bucket of every step + the BN state went through all_reduce
        assert tr.max_over_ranks(1.5) == 1.5
        # the default comm mode is 'auto': at one rank there is no contest (RCCL runs no kernel for a one-rank all-reduce), the
        # overlapped form 'wg' (collectives on the weight-gradient stream) is kept and the report says why
        assert not tr.calibrating and tr.comm_mode == 'wg', (tr.calibrating, tr.comm_mode)
        assert tr.auto_report['chosen'] == 'wg' and 'skipped' in tr.auto_report, tr.auto_report
        assert torch.isfinite(eng.params).all()
        # Stream ordering of every comm mode, on hardware: a probe SCALES the bucket by 0.5 (a kernel of its own) at the very
        # point of the stream where the collective is enqueued -- on the weight-gradient stream for 'wg', the compute stream for
        # 'main' / 'pg', the trainer's stream for 'side'.  A scaling that ran before the range's weight-gradient kernels had
        # finished would leave part of the gradient unscaled.  The slice weight 0.5 goes through fv_train_step's loss_weight
        # (the gradients arrive pre-scaled): together 0.25 x the plain gradient.  1 MiB buckets: the head's range (221 KB) and
        # every small layer's are flushed on their own, right behind their own weight-gradient (ADVICE r3: the head's used to
        # run on the compute stream).
        for cm, bb in (('wg', 8 << 20), ('wg', 1 << 20), ('wg', 64 << 10), ('main', 8 << 20), ('pg', 8 << 20), ('side', 8 << 20)):
            eng.init_synthetic(seed=7)
            eng.iterations = 0; eng.m = eng.v = eng.grads = None
            t2 = DataParallelTrainer(eng, world_size=1, rank=0, bucket_bytes=bb, force_bucket_path=True, comm_mode=cm)
            t2._probe = lambda view: view.mul_(0.5)
            t2.train_on_batch(x, y, 1e-4, 0.99, 0.99, weight=0.5)
            torch.cuda.synchronize()
            rel = ((eng.grads - 0.25 * res[0][1]).norm() / (0.25 * res[0][1]).norm()).item()
            assert rel <= 1e-5, (cm, bb, rel)
            # a direct forward_backward with a callback on the same context afterwards is stream-ordered again (ADVICE r3)
            seen = []
            eng.init_synthetic(seed=7)                  # the step above moved the weights
            eng.forward_backward(x, y, on_bucket=lambda off, cnt: seen.append((off, cnt)))
            torch.cuda.synchronize()
            assert seen and seen[0][0] + seen[0][1] == eng.n_params and seen[-1][0] == 0
            rel = ((eng.grads - res[0][1]).norm() / res[0][1].norm()).item()
            assert rel <= 1e-5, ('plain after ' + cm, rel)
        # the same check on the three-scale training step (fv_yolov3_train_step reports its ranges through the same protocol)
        from face_vijnana_yolov3_amd.yolov3 import Yolov3
        m3 = Yolov3(0, out_channels=18)
        g3 = torch.Generator().manual_seed(5)
        x3 = torch.rand((2, 96, 96, 3), generator=g3).cuda()
        t3 = [torch.rand((2, 96 // d, 96 // d, 18), generator=g3).cuda() for d in (32, 16, 8)]
        m3.init_synthetic(3); m3.ensure_optimizer()
        m3.forward_backward(x3, t3)
        torch.cuda.synchronize()
        g_plain = m3.grads.clone()
        for cm in ('wg', 'main', 'pg'):
            m3.init_synthetic(3); m3.iterations = 0; m3.grads = m3.m = m3.v = None
            t4 = DataParallelTrainer(m3, world_size=1, rank=0, bucket_bytes=1 << 20, force_bucket_path=True, comm_mode=cm)
            t4._probe = lambda view: view.mul_(0.5)
            t4.train_on_batch(x3, t3, 1e-4, 0.9, 0.999, weight=0.5)
            torch.cuda.synchronize()
            rel = ((m3.grads - 0.25 * g_plain).norm() / (0.25 * g_plain).norm()).item()
            assert rel <= 1e-5 and t4.collectives_launched >= 3, ('three-scale', cm, rel, t4.collectives_launched)
        tr.barrier()
        dist.destroy_process_group()
# first step: same loss, same (all-reduced) gradient and first Adam moment up to the float-atomic order inside dW
assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[0][0]), (res[0][0], res[1][0])
for k in (1, 2):
    rel = ((res[0][k] - res[1][k]).norm() / res[0][k].norm()).item()
    assert rel <= 1e-5, (k, rel)
# later steps: Adam's step is lr * m / (sqrt(v) + eps), i.e. sign-like where the gradient is tiny, so rounding-level gradient
# differences move individual parameters by up to 2 lr -- the trajectories stay close, not identical
assert abs(res[0][3] - res[1][3]) <= 5e-3 * abs(res[0][3]), (res[0][3], res[1][3])
torch.testing.assert_close(res[0][4], res[1][4], rtol=5e-3, atol=1e-4)
print('RCCL_WORLD1_OK collectives=%%d backend=nccl' %% res[1][5])
'''


def test_world_size_one_nccl_group_runs_the_bucketed_all_reduce():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run([sys.executable, '-c', WORKER % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert 'RCCL_WORLD1_OK' in r.stdout


def test_bench_through_its_own_launcher_on_one_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--spawn', '--steps', '3', '--warmup', '1', '--batch', '8',
                        '--image-size', '224', '--no-cpu-baseline', '--no-detect', '--no-loader', '--no-three-scale', '--profile-steps', '1'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['value'] > 0 and d['median_ms_per_step'] > 0
    assert d['multi_gpu'] and d['multi_gpu'].get('rccl_ranks') == 1 and d['multi_gpu']['backend'] == 'nccl', d['multi_gpu']
    assert d['multi_gpu']['collectives_per_step'] >= 5
    assert d['multi_gpu']['auto']['chosen'] in ('wg', 'main')
