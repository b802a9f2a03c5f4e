"""SURVEY 8f row 4 end to end: the three-scale head selected through the reference's configuration surface
(nn_arch.head = 'three_scale'), trained from UCCS-format csv rows by FaceDetector.train() (targets: data.encode_gt_three_scale,
loss: fv_yolov3_train_step's objectness / box / class loss, data parallel through the same DataParallelTrainer) and read out by
detect() through the reference's decode_netout / correct_yolo_boxes / do_nms chain (fv_yolo_decode_nms); and the 2-rank
rehearsal of its gradient all-reduce (gloo transport on the one GPU, as tests/test_dp_rehearsal_gpu.py does for the base step)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _conf(root, mode, image_size=96, batch=2):
    return {'mode': mode, 'raw_data_path': root, 'test_path': root, 'output_file_path': os.path.join(root, 'solution_fd.csv'),
            'multi_gpu': False, 'num_gpus': 1, 'yolov3_base_model_load': False, 'model_loading': False, 'bn_zero_debias': False,
            'hps': {'lr': 1e-3, 'beta_1': 0.9, 'beta_2': 0.999, 'decay': 0.0, 'epochs': 1, 'step': 1, 'batch_size': batch,
                    'face_conf_th': 0.5, 'nms_iou_th': 0.5, 'num_cands': 60, 'face_region_ratio_th': 0.8, 'log_every': 5},
            'nn_arch': {'image_size': image_size, 'bb_info_c_size': 6, 'head': 'three_scale', 'num_classes': 1}}


def test_three_scale_train_then_detect_on_synthetic_uccs(tmp_path, monkeypatch):
    """A tiny over-fit: 4 synthetic UCCS images, 160 one-batch epochs at 96x96.  The loss must fall, the saved model must load
    back bit for bit, and detect() on a training image must return the face the csv names: objectness learnt at the assigned
    (cell, anchor), box decoded by the reference's chain to within a few pixels of the letterboxed ground truth."""
    import torch
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.face_detection import FaceDetector
    from face_vijnana_yolov3_amd.postproc import letterbox_device
    monkeypatch.chdir(tmp_path)
    root = str(tmp_path / 'train'); os.makedirs(root)
    rng = np.random.default_rng(5)
    from PIL import Image
    import pandas as pd
    rows = []
    for k in range(4):
        h, w = [(240, 320), (320, 240), (300, 300), (200, 360)][k]
        img = rng.integers(0, 64, (h, w, 3), dtype=np.uint8)
        fw, fh = int(w * 0.3), int(h * 0.45)
        fx, fy = int(rng.integers(10, w - fw - 10)), int(rng.integers(10, h - fh - 10))
        img[fy:fy + fh, fx:fx + fw] = rng.integers(160, 256, (fh, fw, 3), dtype=np.uint8)       # a bright "face"
        name = 'img_%d.jpg' % k
        Image.fromarray(img).save(os.path.join(root, name), quality=95)
        rows.append([k, name, 1, float(fx), float(fy), float(fw), float(fh)])
    pd.DataFrame(rows, columns=data.CSV_COLUMNS).to_csv(os.path.join(root, 'training.csv'), index=False)
    # all four images in ONE batch: the batch statistics then only move with the weights, and with a decaying step the moving
    # statistics (Keras' zero-debiased average, the FaceDetector default) settle on what inference-mode BatchNorm needs
    conf = _conf(root, 'train', batch=4)
    conf['hps'].update(epochs=160, decay=0.03)
    del conf['bn_zero_debias']
    fd = FaceDetector(conf)
    assert fd.three_scale and fd.model.out_channels == 18
    losses = []
    orig = fd.model.forward_backward

    def spy(x, t, on_bucket=None):
        out = orig(x, t, on_bucket=on_bucket)
        if len(losses) % 10 == 0 or len(losses) >= 155:
            losses.append(float(out.item()))
        else:
            losses.append(None)
        return out
    monkeypatch.setattr(fd.model, 'forward_backward', spy)
    fd.train()
    assert fd.model.iterations == 160 and np.isfinite(losses[-1])
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
    assert os.path.exists(FaceDetector.MODEL_PATH)
    conf2 = _conf(root, 'test'); conf2['model_loading'] = True
    fd2 = FaceDetector(conf2)
    assert torch.equal(fd2.model.params, fd.model.params) and torch.equal(fd2.model.state, fd.model.state)
    # detect on the training images (inference-mode BN with the moving statistics of the 160 steps)
    hits = 0
    for k, r in enumerate(rows):
        raw = data._pil_loader(os.path.join(root, r[1]))
        x, geom = letterbox_device(fd2.model.ctx, raw, 96)
        boxes = fd2.detect(x[None])
        h, w = raw.shape[:2]
        tg = data.encode_gt_three_scale([r[3:7]], h, w, 96)
        m = max(h, w)
        _, _, pad_t, _, pad_l, _ = data.letterbox_geometry(h, w, 96)
        ox, oy = (0, pad_t) if w >= h else (pad_l, 0)
        gx1, gy1 = r[3] / m * 96 + ox, r[4] / m * 96 + oy
        gx2, gy2 = (r[3] + r[5]) / m * 96 + ox, (r[4] + r[6]) / m * 96 + oy
        assert sum(int((t.reshape(t.shape[0], t.shape[1], 3, 6)[..., 4] == 1).sum()) for t in tg) == 1
        for b in boxes:
            ix = max(0.0, min(b.xmax, gx2) - max(b.xmin, gx1)) * max(0.0, min(b.ymax, gy2) - max(b.ymin, gy1))
            un = (b.xmax - b.xmin) * (b.ymax - b.ymin) + (gx2 - gx1) * (gy2 - gy1) - ix
            if un > 0 and ix / un >= 0.5 and b.get_score() >= 0.5:
                hits += 1
                break
    assert hits >= 3, hits                 # the over-fitted detector finds (at least) three of its four training faces
    # test(): the reference's csv surface over the same folder
    fd2.conf = dict(conf2, test_path=root, output_file_path=os.path.join(root, 'solution_3s.csv'))
    fd2.test()
    out = [l.strip().split(',') for l in open(os.path.join(root, 'solution_3s.csv'))]
    assert out and all(len(r) == 6 for r in out)


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from face_vijnana_yolov3_amd.yolov3 import Yolov3
from face_vijnana_yolov3_amd.parallel import DataParallelTrainer, slice_batch
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
out = %(out)r
m = Yolov3(0, out_channels=18)
m.init_synthetic(seed=7 + rank)                   # deliberately different: the trainer must broadcast rank 0's
tr = DataParallelTrainer(m, world_size=world, rank=rank, bucket_bytes=32 << 20)
N, S = 5, 64
g = torch.Generator().manual_seed(100)            # the GLOBAL batch, identical on every rank
x_all = torch.rand((N, S, S, 3), generator=g)
t_all = []
for d in (32, 16, 8):
    t = torch.rand((N, S // d, S // d, 18), generator=g)
    t4 = t.view(N, S // d, S // d, 3, 6); t4[..., 4:] = (t4[..., 4:] > 0.8).float()
    t_all.append(t)
lo, hi, weight = slice_batch(N, world, rank)      # 2 + 3 images
x = x_all[lo:hi].cuda(); tg = [t[lo:hi].cuda() for t in t_all]
if rank == 0:
    np.save(os.path.join(out, 'p0.npy'), m.params.cpu().numpy()); np.save(os.path.join(out, 's0.npy'), m.state.cpu().numpy())
loss = tr.train_on_batch(x, tg, 1e-4, 0.99, 0.99, weight=weight)
torch.cuda.synchronize()
merged = tr.merged_loss(loss, weight)
pos = m.leaky_slopes_taken(hi - lo, S)
np.save(os.path.join(out, 'pos%%d.npy' %% rank), np.packbits(torch.cat([q.reshape(-1) for q in pos]).cpu().numpy()))
cover = sorted(tr.reducer.launched)
ok_cover = cover[0][0] == 0 and cover[-1][1] == m.n_params and all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))
np.savez(os.path.join(out, 'rank%%d.npz' %% rank), grads=m.grads.cpu().numpy(), params=m.params.cpu().numpy(), state=m.state.cpu().numpy(),
         loss=float(loss.item()), merged=merged, ok_cover=ok_cover, nbuckets=len(cover), lo=lo, hi=hi)
tr.shutdown()
'''


def test_three_scale_two_ranks_match_the_merged_batch_oracle(tmp_path):
    import torch
    from oracle import net_oracle as no
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, FV_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    port = 29900 + os.getpid() % 90
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    a = np.load(tmp_path / 'rank0.npz'); b = np.load(tmp_path / 'rank1.npz')
    assert a['ok_cover'] and b['ok_cover'] and a['nbuckets'] >= 5
    assert np.array_equal(a['grads'], b['grads']) and np.array_equal(a['params'], b['params']) and np.array_equal(a['state'], b['state'])
    assert float(a['merged']) == float(b['merged']) and float(a['loss']) != float(b['loss'])
    N, S, out_ch = 5, 64, 18
    ents, n_params, _ = no.yolov3_layout(out_ch)
    p0 = torch.from_numpy(np.load(tmp_path / 'p0.npy')); s0 = torch.from_numpy(np.load(tmp_path / 's0.npy'))
    g = torch.Generator().manual_seed(100)
    x_all = torch.rand((N, S, S, 3), generator=g)
    t_all = []
    for d in (32, 16, 8):
        t = torch.rand((N, S // d, S // d, 18), generator=g)
        t4 = t.view(N, S // d, S // d, 3, 6); t4[..., 4:] = (t4[..., 4:] > 0.8).float()
        t_all.append(t)
    from face_vijnana_yolov3_amd.yolov3 import yolov3_layer_table
    layers = [d for d in yolov3_layer_table(out_ch) if d['has_bn']]
    g64 = torch.zeros(n_params, dtype=torch.float64); g32 = torch.zeros(n_params, dtype=torch.float64)
    l64 = 0.0
    for rk, (lo, hi) in enumerate(((0, 2), (2, 5))):
        bits = torch.from_numpy(np.unpackbits(np.load(tmp_path / ('pos%d.npy' % rk)))).bool()
        pos, o = [], 0
        for d in layers:
            shape = (hi - lo, S // d['out_div'], S // d['out_div'], d['cout'])
            cnt = int(np.prod(shape))
            pos.append(bits[o:o + cnt].view(shape)); o += cnt
        w = (hi - lo) / float(N)
        xs = x_all[lo:hi]; ts = [t[lo:hi] for t in t_all]
        l, gr, _ = no.yolov3_train_step_grads(p0.double(), s0.double(), xs.double(), [t.double() for t in ts], out_ch, positive=pos)
        _, gr32, _ = no.yolov3_train_step_grads(p0, s0, xs, ts, out_ch, positive=pos)
        g64 += w * gr; g32 += w * gr32.double(); l64 += w * float(l)
    assert abs(float(a['merged']) - l64) <= 1e-5 * abs(l64)
    got = torch.from_numpy(a['grads']).double()
    worst = 0.0
    for e in ents:
        k, cin, cout = e['k'], e['cin'], e['cout']
        parts = [slice(e['w_off'], e['w_off'] + cout * k * k * cin)]
        parts += [slice(e[nm], e[nm] + cout) for nm in ('gamma_off', 'beta_off')] if e['has_bn'] else [slice(e['bias_off'], e['bias_off'] + cout)]
        for sl in parts:
            n64 = g64[sl].norm().item()
            rel = (got[sl] - g64[sl]).norm().item() / max(n64, 1e-30)
            rel32 = (g32[sl] - g64[sl]).norm().item() / max(n64, 1e-30)
            assert rel <= max(6 * rel32, 4e-5), (e['name'], rel, rel32)
            worst = max(worst, rel)
    print('three-scale 2-rank reduced gradient vs merged-batch oracle: worst rel-L2 %.2e' % worst)
