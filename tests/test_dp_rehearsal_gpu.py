"""N>1 on real device tensors: two ranks on the one GPU of the test box, gloo as the transport
(RCCL refuses two ranks on one device: "Duplicate GPU detected"), driving the same
DataParallelTrainer code the RCCL path uses -- bucket callbacks from fv_train_step, side-stream
all-reduce, BN-state averaging, redundant Adam."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from face_vijnana_yolov3_amd.engine import Engine
from face_vijnana_yolov3_amd.parallel import DataParallelTrainer
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
eng = Engine(0)
eng.init_synthetic(seed=7 + rank)                 # deliberately different: the trainer must broadcast rank 0's
tr = DataParallelTrainer(eng, world_size=world, rank=rank, bucket_bytes=16 << 20)
g = torch.Generator().manual_seed(100 + rank)     # every rank its own slice of the global batch
x = torch.rand((3, 96, 96, 3), generator=g).cuda(); y = torch.rand((3, 3, 3, 6), generator=g).cuda()
losses = [float(tr.train_on_batch(x, y, 1e-4, 0.99, 0.99).item()) for _ in range(3)]
torch.cuda.synchronize()
cover = sorted(tr.reducer.launched)
ok_cover = cover[0][0] == 0 and cover[-1][1] == eng.n_params and all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))
np.savez(os.path.join(%(out)r, 'rank%%d.npz' %% rank), params=eng.params.cpu().numpy(), state=eng.state.cpu().numpy(),
         losses=np.array(losses), ok_cover=ok_cover, nbuckets=len(cover), iterations=eng.iterations)
tr.shutdown()
'''


def test_two_ranks_stay_bit_identical(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, FV_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    a = np.load(tmp_path / 'rank0.npz'); b = np.load(tmp_path / 'rank1.npz')
    assert a['ok_cover'] and b['ok_cover'] and a['nbuckets'] >= 5 and a['iterations'] == 3
    # identical start (broadcast), identical averaged gradients and BN state -> bit-identical replicas
    assert np.array_equal(a['params'], b['params'])
    assert np.array_equal(a['state'], b['state'])
    assert np.isfinite(a['losses']).all() and np.isfinite(b['losses']).all()
    assert not np.array_equal(a['losses'], b['losses'])          # each rank reports its own slice's loss
