#!/usr/bin/env python3
"""Headline benchmark: FaceDetector training images/sec at 416x416, 40 images per GPU
(BASELINE.json configs[1]; N GPUs -> global batch 40*N, weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = forward (training-mode BN) + MSE + backward + (N>1: RCCL all-reduce of the 40.64 M fp32
gradients, bucketed and overlapped with backward) + Keras-formula Adam, on synthetic inputs that
are already resident in HBM.  Rank 0 prints ONE JSON line.

`roofline`: the dominant kernel FAMILY is the 128x128-tile fp32-MFMA implicit-GEMM conv -- forward and data-gradient of every layer
with >= 128 output channels, 92 launches per step: conv_kernel<128,2,4,false> (8 waves per workgroup; <128,2,2,false> under option
"conv_waves8" = 0) and, since round 5, conv1x1_persist_kernel<128> for its 1x1 launches with more than 512 tiles (the same tile, K loop
and epilogue, persistent over tiles).  achieved = algorithmic FLOPs of the family's launches / their HIP-event-timed duration, taken in
instrumented steps right after the timed region (the timed steps themselves run un-instrumented); `parts` gives each kernel name.
peak = 157.3 TFLOP/s, the dense fp32 MFMA rate of MI355X (MI355X_MICROARCH.md).
`cpu_baseline`: the torch-CPU oracle restatement of the same step (kind "port"; the Keras/TF
reference cannot run here) on a bounded sample, timed on this box's host cores; `detect_cpu` inside it is
the detect path on the CPU (oracle forward at batch 1 + the single-core C oracle of decode/NMS).
`loader_inclusive`: the same step fed by FaceDetector.train's real input path (JPEG decode on host
threads, one pinned H2D copy and one fv_letterbox_batch launch per batch, GT encoding) from a synthetic
UCCS-format folder -- reported beside `value`, never as `value`.
`three_scale_train`: one training step of the full three-scale YOLOv3 graph at batch 16 -- secondary.
N > 1: `multi_gpu` carries per-rank all-reduce time, the exposed communication (step minus a
compute-only step) and the RCCL world size, so that a scaling run diagnoses itself.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')   # before the HIP runtime initialises (face_vijnana_yolov3_amd/__init__.py says why); inherited by the ranks

IMAGE_SIZE = 416
PER_GPU_BATCH = 40
FP32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (about 6.3 TB/s is what a float4 copy reaches)
# The dominant kernel FAMILY: the 128x128-tile fp32-MFMA implicit GEMM of every forward / data-gradient launch with >= 128 output
# channels.  Since round 5 its 1x1 launches with more than 512 tiles run as conv1x1_persist_kernel<128,...> (the same tile, K loop
# and epilogue, persistent over tiles; conv1x1_mfma.hip), so the family is two kernel names: the roofline is computed over BOTH
# (the same 92 launches per step as in rounds 1-4) and `parts` gives each name on its own.
DOMINANT_RE = r'conv_kernel<128, ?2, ?[24], ?false|conv1x1_persist_kernel<128'   # library labels / rocprofv3 names


def dominant_parts(d):
    """{name: record} of the dominant family's kernels in a per-kernel dict."""
    import re
    return {k: v for k, v in d.items() if re.match(DOMINANT_RE, k)}


def dominant(d):
    """(name, record) of the dominant family in a per-kernel profile dict: launches, ms, flops and bytes summed over its kernels."""
    parts = dominant_parts(d)
    if not parts:
        return None, None
    tot = {f: sum(v[f] for v in parts.values()) for f in ('launches', 'ms', 'flops', 'bytes')}
    return ' + '.join(sorted(parts)), tot


HPS = dict(lr=1e-4, beta_1=0.99, beta_2=0.99, decay=0.0)  # reference face_vijnana_yolov3.json:12-15
TRAFFIC_FILES = ('r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--spawn', action='store_true', help='go through the self-launcher even for --gpus 1 (rehearsal of the N>1 start-up)')
    ap.add_argument('--rendezvous-only', action='store_true', help='ranks only rendezvous, all-reduce one number and exit (launcher rehearsal)')
    ap.add_argument('--batch', type=int, default=PER_GPU_BATCH, help='per-GPU batch (default 40 = BASELINE config)')
    ap.add_argument('--image-size', type=int, default=IMAGE_SIZE)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--profile-steps', type=int, default=2)
    ap.add_argument('--no-overlap', action='store_true', help='serialise wgrad on the main stream (A/B aid)')
    ap.add_argument('--no-tail-split', action='store_true', help='conv launches without the tail split (A/B aid)')
    ap.add_argument('--no-detect', action='store_true', help='skip the detect-path measurement (PMC passes)')
    ap.add_argument('--no-rccl-rehearsal', action='store_true', help='N=1: skip the world-size-1 RCCL group rehearsal')
    ap.add_argument('--no-loader', action='store_true', help='skip the loader-inclusive measurement')
    ap.add_argument('--no-three-scale', action='store_true', help='skip the three-scale training measurement')
    ap.add_argument('--loader-steps', type=int, default=16)
    return ap.parse_args()


def host_cpu():
    """CPU model string, physical cores and logical CPUs of this box (from /proc/cpuinfo)."""
    model, cores = 'unknown', set()
    try:
        phys = core = None
        for ln in open('/proc/cpuinfo'):
            k, _, v = ln.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'model name':
                model = v
            elif k == 'physical id':
                phys = v
            elif k == 'core id':
                core = v
            elif not k and phys is not None:
                cores.add((phys, core)); phys = core = None
    except OSError:
        pass
    return model, len(cores) or None, os.cpu_count()


def cpu_baseline(batch, image_size, budget_s=45.0):
    """Torch-CPU oracle (the reference's arithmetic restated, kind 'port') on this box's host cores, on a
    bounded sample: train step (fwd+bwd+Adam) at `batch` images, warm-up + timed samples; plus the detect
    path on the CPU: oracle forward at batch 1 and the single-core C oracle of decode + NMS + top-k."""
    import numpy as np
    import torch
    from oracle import net_oracle as no
    from oracle import postproc as opp
    torch.manual_seed(0)
    p, st = no.init_params(7, torch.float32)
    g = image_size // 32
    m = torch.zeros_like(p); v = torch.zeros_like(p)

    def step(b, s):
        x = torch.rand((b, s, s, 3)); yt = torch.rand((b, s // 32, s // 32, 6))
        t0 = time.perf_counter()
        loss, grad, _ = no.train_step_grads(p, st, x, yt)
        no.keras_adam(p, grad, m, v, 0, **{k: HPS[k] for k in ('lr', 'beta_1', 'beta_2')})
        return time.perf_counter() - t0

    # thread sweep (128 threads at batch 1 ran the oracle at 10 GFLOP/s): the step at `batch` images with
    # 16 / 32 / 64 torch threads (capped at the box's logical CPUs), best kept; bounded -- the sweep stops once `budget_s`
    # seconds of timed CPU work have been spent
    logical = os.cpu_count() or 8
    sweep = sorted({min(t, logical) for t in (16, 32, 64)})   # (a 1-GPU box of this pool owns a 16-core share of its host; 128 threads: 20-21 s, 3x the best)
    t_by_threads, spent = {}, 0.0
    for nt in sweep:
        torch.set_num_threads(nt)
        step(1, 128)                               # warm-up: thread pool, allocator, oneDNN primitives
        t = step(batch, image_size)
        t_by_threads[nt] = t; spent += t
        if spent > budget_s:
            break
    best_threads = min(t_by_threads, key=t_by_threads.get)
    torch.set_num_threads(best_threads)
    t_train = [t_by_threads[best_threads]]
    if spent < budget_s:
        t_train.append(step(batch, image_size))
    x1 = torch.rand((1, image_size, image_size, 3))
    with torch.no_grad():
        no.forward(p, st, x1, training=False)
        t_fwd = []
        for _ in range(3):
            t0 = time.perf_counter(); no.forward(p, st, x1, training=False); t_fwd.append(time.perf_counter() - t0)
    rng = np.random.default_rng(99)
    nf = 10000
    head = np.zeros((nf, g, g, 6), np.float32)
    head[..., 0] = rng.normal(0, 2, (nf, g, g)); head[..., 5] = rng.normal(0, 2, (nf, g, g))
    head[..., 1:3] = rng.uniform(0, 1, (nf, g, g, 2)); head[..., 3:5] = rng.uniform(0, 0.3, (nf, g, g, 2))
    opp.detect_postproc(head[:64], image_size, 0.5, 0.5, 60)
    t0 = time.perf_counter(); opp.detect_postproc(head, image_size, 0.5, 0.5, 60); t_pp = time.perf_counter() - t0
    model, phys, logical = host_cpu()
    best = min(t_train)
    train_flops = 3 * no.fwd_flops_per_image(image_size) - 2 * image_size * image_size * 27 * 32
    return dict(value=round(batch / best, 4), unit='images/sec', cores=torch.get_num_threads(), kind='port',
                cpu_model=model, physical_cores=phys, logical_cpus=logical,
                samples_s=[round(t, 2) for t in t_train], thread_sweep_s={str(k): round(v, 2) for k, v in t_by_threads.items()},
                gflops=round(train_flops * batch / best / 1e9, 1),
                sample='train step (fwd+bwd+Adam) of the torch-CPU oracle at batch %d, %dx%d, fp32: thread sweep %s (one timed step each '
                       'after a 128x128 warm-up, bounded at %.0f s), best = %d threads, %d timed samples there, best taken'
                       % (batch, image_size, image_size, sorted(t_by_threads), budget_s, best_threads, len(t_train)),
                detect_cpu=dict(unit='ms/img', forward_batch1=round(min(t_fwd) * 1e3, 1), forward_threads=torch.get_num_threads(),
                                postproc_c_oracle_1core=round(t_pp / nf * 1e3, 5),
                                sample='oracle forward (inference BN) at batch 1, best of 3; C oracle decode+NMS+top-k over the %d '
                                       'config-4 frames on one core' % nf))


def detect_bench(eng, x40):
    """BASELINE metric 2, detect-path ms/img (fd.py:885-949 = predict + decode/NMS/top-k):
    batch 1 as the reference's evaluate loop calls it, batch 40, and config 4 (post-processing alone
    on 10k synthetic head outputs).  Device-side times (stream-ordered, one sync at the end)."""
    import numpy as np
    import torch
    from face_vijnana_yolov3_amd.postproc import decode_nms, to_boundboxes

    def timed(fn, reps):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    x1 = x40[:1].contiguous()
    S = x40.shape[1]
    one = lambda xb: decode_nms(eng.ctx, eng.predict_device(xb), S, 0.5, 0.5, 60)
    t1 = timed(lambda: one(x1), 20)
    t40 = timed(lambda: one(x40), 5)
    t0 = time.perf_counter()
    for _ in range(20):
        to_boundboxes(one(x1), 0)           # end to end incl. D2H + BoundBox objects, as detect() returns
    e2e = (time.perf_counter() - t0) / 20 * 1e3
    rng = np.random.default_rng(99)
    g = S // 32
    head = np.zeros((10000, g, g, 6), np.float32)
    head[..., 0] = rng.normal(0, 2, (10000, g, g)); head[..., 5] = rng.normal(0, 2, (10000, g, g))
    head[..., 1:3] = rng.uniform(0, 1, (10000, g, g, 2)); head[..., 3:5] = rng.uniform(0, 0.3, (10000, g, g, 2))
    hd = torch.from_numpy(head).cuda()
    tpp = timed(lambda: decode_nms(eng.ctx, hd, S, 0.5, 0.5, 60), 10)
    out = dict(unit='ms/img', batch1_device=round(t1, 4), batch1_end_to_end=round(e2e, 4),
               batch40_device=round(t40 / x40.shape[0], 4), postproc_10k_frames=round(tpp / 10000, 6),
               postproc_10k_total_ms=round(tpp, 3))
    return out


def loader_bench(eng, trainer, B, S, steps):
    """SURVEY 8d config 2 'data-loader-inclusive number': FaceDetector.train's input path + the step."""
    import numpy as np
    import torch
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.face_detection import BatchFeeder, run_pipelined
    with tempfile.TemporaryDirectory() as root:
        n_img = 2 * B
        rng = np.random.default_rng(0)
        sizes = [(768, 1024), (1024, 768), (720, 1280), (600, 800)]
        from PIL import Image
        rows, fid = [], 0
        for k in range(n_img):
            h, w = sizes[k % len(sizes)]
            lo = rng.integers(0, 256, (h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)     # photo-like spectrum, not white noise
            im = Image.fromarray(lo).resize((w, h), Image.BICUBIC)
            name = 'img_%04d.jpg' % k
            im.save(os.path.join(root, name), quality=90)
            for _ in range(int(rng.integers(1, 6))):
                fw = float(rng.uniform(20, w / 4)); fh = float(rng.uniform(20, h / 4))
                rows.append([fid, name, 1, round(float(rng.uniform(1, w - fw - 1)), 1), round(float(rng.uniform(1, h - fh - 1)), 1), round(fw, 1), round(fh, 1)])
                fid += 1
        import pandas as pd
        pd.DataFrame(rows, columns=data.CSV_COLUMNS).to_csv(os.path.join(root, 'training.csv'), index=False)
        hps = dict(HPS, batch_size=B, step=1)
        seq = data.TrainingSequence(root, hps, {'image_size': S, 'bb_info_c_size': 6})
        threads = min(16, max(2, (os.cpu_count() or 8) // 2))
        def loader_rate(f):
            t0 = time.perf_counter()
            for k in range(3):
                f.load(k % len(seq))
            return 3 * B / (time.perf_counter() - t0)
        # loader alone, both decoders, interleaved A/B/A/B after a warm-up of each (file cache, three pinned staging buffers each);
        # best of two: the host share of a 1-GPU box is noisy
        pil = BatchFeeder(data.TrainingSequence(root, dict(hps, device_jpeg=False), {'image_size': S, 'bb_info_c_size': 6}), 1, 0, threads)
        feeder = BatchFeeder(seq, 1, 0, threads)       # default: Huffman decoding on the host, the rest of the JPEG decode on the device
        loader_rate(pil); loader_rate(feeder)
        loader_only_pillow = loader_only = 0.0
        for _ in range(2):
            loader_only_pillow = max(loader_only_pillow, loader_rate(pil))
            loader_only = max(loader_only, loader_rate(feeder))
        pil.close()
        # ONE pipelined run of 4 + steps batches; steady state = from the end of step 3 to the end of the last step, between two
        # events on the compute stream (after_step(k) is called once step k + 1 is in the queue).  The run's first batch is
        # decoded and staged with nothing to overlap it (~30 ms of pipeline fill, once per epoch): reported beside, not inside.
        n = 4 + steps
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        def after_step(k, loss, item):
            if k == 2:
                ev0.record()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_pipelined(eng, trainer, feeder, [k % len(seq) for k in range(n)], S, hps, after_step)
        ev1.record()
        torch.cuda.synchronize()
        whole = (time.perf_counter() - t0) / n * 1e3
        dt = ev0.elapsed_time(ev1) * 1e-3
        feeder.close()
    return dict(value=round(B * steps / dt, 2), unit='images/sec', ms_per_step=round(dt / steps * 1e3, 3), steps=steps,
                ms_per_step_with_pipeline_fill=round(whole, 3),
                loader_only_images_per_sec=round(loader_only, 1), loader_only_pillow_images_per_sec=round(loader_only_pillow, 1),
                loader_threads=threads,
                path='%d synthetic UCCS-format JPEGs (768x1024 .. 720x1280): Huffman decoding on host threads (fv_jpeg_entropy_decode) '
                     '-> quantised coefficients in one pinned buffer -> H2D -> fv_jpeg_reconstruct_batch (IDCT, chroma upsampling, '
                     'colour conversion on the device) -> fv_letterbox_batch on a staging stream -> fv_train_step + Adam; batch k+2 decoded '
                     'and batch k+1 staged while step k runs; loader_only_pillow = the same loader with the whole decode in Pillow on the host' % n_img)


def test_loop_bench(device, S, n_img=512, head='single'):
    """FaceDetector.test() end to end (fd.py:783-883: JPEG decode -> letterbox -> predict -> decode/NMS/top-k -> back-projection
    -> csv rows) on a synthetic UCCS-format folder: images/sec at the reference's batch 1 and with the read-ahead batches of
    hps.eval_batch_size = 16 / 32 / the default (face_detection.default_eval_batch: 48 at 416^2), then at the default for
    hps.loader_threads = 8 / 16 / 32 / 64.  Wall clock, host work included.  n_img: the loop is a two-deep pipeline whose first
    load and last forward overlap nothing -- with 64 images a batch of 32 is two chunks, i.e. all fill and drain (round 4's
    three-scale 709 -> 537 img/s from batch 16 to 32 was that, not the kernels); 512 images = 16 chunks of 32.  The loop is bound by
    the device: `device_only` is network + decode/NMS of a batch alone (HIP events), the main thread spends
    12-13 of its 14-16 ms per batch waiting for that (tools/test_loop_scale_probe.py: 1 971 / 2 074 / 2 255 img/s at 256 / 512 /
    2 048 images against 2 362 device-only)."""
    import numpy as np
    from PIL import Image
    from face_vijnana_yolov3_amd import face_detection
    with tempfile.TemporaryDirectory() as root:
        rng = np.random.default_rng(0)
        sizes = [(768, 1024), (1024, 768), (720, 1280), (600, 800)]
        base = []
        for k in range(16):                                  # 16 distinct pictures, written n_img / 16 times each (file cache alike)
            h, w = sizes[k % len(sizes)]
            lo = rng.integers(0, 256, (h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)
            base.append(Image.fromarray(lo).resize((w, h), Image.BICUBIC))
        for k in range(n_img):
            base[k % 16].save(os.path.join(root, 'img_%04d.jpg' % k), quality=90)
        default_threads = face_detection.default_loader_threads()
        conf = {'mode': 'test', 'raw_data_path': root, 'test_path': root, 'output_file_path': os.path.join(root, 'solution_fd.csv'),
                'multi_gpu': False, 'num_gpus': 1, 'yolov3_base_model_load': False, 'model_loading': False,
                'hps': dict(HPS, epochs=1, step=1, batch_size=40, face_conf_th=0.5, nms_iou_th=0.5, num_cands=60, loader_threads=default_threads),
                'nn_arch': {'image_size': S, 'bb_info_c_size': 6, 'head': head}}
        dbg, face_detection.DEBUG = face_detection.DEBUG, False
        try:
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()):
                fd = face_detection.FaceDetector(conf, device)
            if head == 'single':
                d = fd.model.layers[-1]                       # a head that fires on a few cells, so that rows are written
                fd.model.params[d['w_off']:d['beta_off']] *= 0.05
                fd.model.params[d['beta_off']] = 0.3; fd.model.params[d['beta_off'] + 5] = 0.3
            else:
                for d in fd.model.layers:                     # the three detection convs: objectness logits around -2, a few cells fire
                    if not d['has_bn']:
                        fd.model.params[d['w_off']:d['beta_off']] *= 0.05
                        fd.model.params[d['beta_off'] + 4:d['beta_off'] + d['cout']:6] = -2.0

            def rate():
                fd.test()                                 # warm-up (workspace, pinned ring, file cache)
                best = 0.0
                for _ in range(2):                        # best of two: the host share of a 1-GPU box is noisy (16 CPUs of a 256-thread host)
                    t0 = time.perf_counter(); fd.test(); dt = time.perf_counter() - t0
                    best = max(best, n_img / dt)
                return round(best, 1)
            out = {}
            dflt = face_detection.default_eval_batch(S)
            for bs in ((1, 16, 32, dflt) if head == 'single' else (16, 32, dflt)):
                conf['hps']['eval_batch_size'] = bs
                out['batch%d' % bs] = rate()
            rows = sum(1 for _ in open(conf['output_file_path']))
            # the device's own share of a batch of 32 (network + decode/NMS, events on the compute stream): the loop's ceiling
            import torch
            ceiling = {}
            for bs in (32, dflt):
                xs = torch.rand((bs, S, S, 3), device='cuda')
                for _ in range(2):
                    fd._detect_collect(fd._detect_launch(xs))
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    launched = fd._detect_launch(xs)
                e1.record(); torch.cuda.synchronize()
                ceiling['batch_%d' % bs] = round(bs * 5e3 / e0.elapsed_time(e1), 1)
            sweep = {}
            if head == 'single':
                conf['hps']['eval_batch_size'] = dflt
                for nt in (8, 16, 32, 64):
                    conf['hps']['loader_threads'] = nt
                    sweep[str(nt)] = rate()
        finally:
            face_detection.DEBUG = dbg
    if head != 'single':
        return dict(unit='images/sec', eval_batch_16=out['batch16'], eval_batch_32=out['batch32'], eval_batch_default=dflt,
                    eval_batch_at_default=out['batch%d' % dflt], images=n_img, csv_rows=rows,
                    loader_threads=default_threads, device_only=ceiling,
                    path='FaceDetector.test() with nn_arch.head = three_scale: fv_yolov3_forward + fv_yolo_decode_nms_batch (one launch pair per batch)')
    return dict(unit='images/sec', eval_batch_1=out['batch1'], eval_batch_16=out['batch16'], eval_batch_32=out['batch32'],
                eval_batch_default=dflt, eval_batch_at_default=out['batch%d' % dflt], images=n_img, csv_rows=rows,
                loader_threads=default_threads, eval_batch_default_by_loader_threads=sweep, device_only=ceiling, host_cpus=host_cpus(),
                path='FaceDetector.test(): %d synthetic JPEGs (768x1024 .. 720x1280), Huffman decoding on hps.loader_threads host threads (default '
                     'the usable CPUs, 4 .. 32) into reused pinned buffers one batch ahead, fv_jpeg_reconstruct_batch, fv_letterbox_batch, '
                     'fv_forward_infer, fv_decode_nms, back-projection, csv' % n_img)


def host_cpus():
    """Logical CPUs of the host and the CPUs this process may actually run on (affinity mask / cgroup quota)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None
    quota = None
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        quota = None if q == 'max' else round(int(q) / int(per), 1)
    except (OSError, ValueError):
        pass
    return dict(logical=os.cpu_count(), affinity=aff, cgroup_quota=quota)


def config5_bench(device, B=16, S=608, steps=10):
    """BASELINE config 5 at its per-GPU size: image_size 608, batch 16 (grid 19; the x8 data-parallel half is the driver's).
    Its own Engine (the headline engine keeps its workspace), device-resident synthetic batch, median of `steps` steps."""
    import torch
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.engine import Engine, train_flops_per_image
    eng = Engine(device)
    eng.init_synthetic(seed=7)
    g = torch.Generator(device='cpu').manual_seed(608)
    x = torch.rand((B, S, S, 3), generator=g).cuda(device)
    y = torch.from_numpy(data.synth_gt_batch(B, S, seed=608)).cuda(device)
    for _ in range(3):
        loss = eng.train_on_batch(x, y, **HPS)
    torch.cuda.synchronize(device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        loss = eng.train_on_batch(x, y, **HPS); ev[i + 1].record()
    torch.cuda.synchronize(device)
    per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    ms = per[len(per) // 2] if len(per) % 2 else 0.5 * (per[len(per) // 2 - 1] + per[len(per) // 2])
    tf = train_flops_per_image(S) * B / (ms * 1e-3) / 1e12
    xi = x[:1].contiguous()
    for _ in range(3):
        eng.predict_device(x); eng.predict_device(xi)
    torch.cuda.synchronize(device)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(5):
        eng.predict_device(x)
    e1.record()
    for _ in range(20):
        eng.predict_device(xi)
    e2.record(); torch.cuda.synchronize(device)
    out = dict(value=round(B / (ms * 1e-3), 2), unit='images/sec', median_ms_per_step=round(ms, 3), batch=B, image_size=S, steps=steps,
               step_tflops=round(tf, 2), frac_of_fp32_mfma_peak=round(tf / FP32_MFMA_PEAK_TFLOPS, 4), loss=float(loss.item()),
               forward_infer_ms_per_img_batch16=round(e0.elapsed_time(e1) / 5 / B, 4), forward_infer_ms_per_img_batch1=round(e1.elapsed_time(e2) / 20, 4),
               workload='FaceDetector mode=train image_size=608 batch_size=16 (BASELINE config 5, one GPU of the 16x8): fv_train_step + Adam')
    del eng, x, y
    torch.cuda.empty_cache()
    return out


def three_scale_bench(device, S, B=PER_GPU_BATCH, steps=3):
    """one training step of the full three-scale YOLOv3 graph (75 convs, two upsample+concat routes,
    255 output channels, the build's objectness/box/class loss, Adam) at the headline's per-GPU batch -- device-resident synthetic batch.  A secondary
    number beside `value`; weight-gradients overlap the data-gradient chain on the side stream as in fv_train_step."""
    import torch
    from face_vijnana_yolov3_amd.yolov3 import Yolov3
    m = Yolov3(device, out_channels=255)
    m.init_synthetic(3)
    g = torch.Generator().manual_seed(4)
    x = torch.rand((B, S, S, 3), generator=g).cuda(device)
    tg = []
    for d in (32, 16, 8):
        t = torch.rand((B, S // d, S // d, 255), generator=g)
        t4 = t.view(B, S // d, S // d, 3, 85)
        t4[..., 4] = (t4[..., 4] > 0.9).float(); t4[..., 5:] = (t4[..., 5:] > 0.98).float()
        tg.append(t.cuda(device))
    for _ in range(2):
        loss = m.train_on_batch(x, tg, HPS['lr'], HPS['beta_1'], HPS['beta_2'], HPS['decay'])
    torch.cuda.synchronize(device)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        loss = m.train_on_batch(x, tg, HPS['lr'], HPS['beta_1'], HPS['beta_2'], HPS['decay'])
    e1.record(); torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / steps
    tf = m.train_flops_per_image(S) * B / (ms * 1e-3) / 1e12
    out = dict(value=round(B / (ms * 1e-3), 1), unit='images/sec', ms_per_step=round(ms, 2), batch=B, image_size=S, steps=steps,
               step_tflops=round(tf, 1), frac_of_fp32_mfma_peak=round(tf / FP32_MFMA_PEAK_TFLOPS, 4), loss=float(loss.item()),
               workload='fv_yolov3_train_step + Adam: make_yolov3_model graph (yd.py:217-311), 255 output channels, synthetic targets')
    del m
    torch.cuda.empty_cache()
    return out


def rccl_world1_rehearsal(eng, x, y, steps=10):
    """N = 1 only: the data-parallel step with a REAL RCCL process group of one rank -- every gradient bucket and the BN state
    go through dist.all_reduce (backend nccl = RCCL) on the communication stream, ordered by events against the backward pass,
    exactly as on 8 GPUs.  Reported in `multi_gpu` so that the N = 1 line already shows the collective path alive."""
    import torch
    import torch.distributed as dist
    from face_vijnana_yolov3_amd.parallel import DataParallelTrainer
    if dist.is_initialized():
        return None
    try:
        # an in-process store: under torch.distributed.run a tcp:// rendezvous would try to join the elastic agent's store
        def timed(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3
        plain_ms = timed(lambda: eng.train_on_batch(x, y, **HPS))           # the same loop without the DP machinery, for comparison
        dist.init_process_group('nccl', store=dist.HashStore(), rank=0, world_size=1, device_id=eng.dev)
        tr = DataParallelTrainer(eng, world_size=1, rank=0, force_bucket_path=True, comm_mode='wg')
        ms = timed(lambda: tr.train_on_batch(x, y, **HPS))
        tr_auto = DataParallelTrainer(eng, world_size=1, rank=0, force_bucket_path=True, comm_mode='auto')
        for _ in range(32):                     # the default mode: measures 'wg' and 'main' over its first steps, keeps the faster
            if not tr_auto.calibrating:
                break
            tr_auto.train_on_batch(x, y, **HPS)
        tr_main = DataParallelTrainer(eng, world_size=1, rank=0, force_bucket_path=True, comm_mode='main')
        main_ms = timed(lambda: tr_main.train_on_batch(x, y, **HPS))
        n_before = tr.collectives_launched
        tr.train_on_batch(x, y, **HPS)
        n_coll = tr.collectives_launched - n_before
        torch.cuda.synchronize()
        ar = tr.allreduce_ms()                  # the same collectives back to back, nothing overlapping them
        out = dict(rccl_ranks=dist.get_world_size(), backend=dist.get_backend(), bucket_mib=tr.bucket_bytes >> 20,
                   gradient_mb=round(eng.n_params * 4 / 1e6, 2), collectives_per_step=n_coll, allreduce_ms=round(ar, 3), comm_mode=tr.comm_mode,
                   auto=tr_auto.auto_report, ms_per_step=round(ms, 3), blocking_on_compute_stream_ms_per_step=round(main_ms, 3),
                   plain_ms_per_step_same_loop=round(plain_ms, 3), steps=steps,
                   note='world-size-1 nccl group on this GPU: bucketed all_reduce calls on the weight-gradient stream, beside the data-gradient chain')
        dist.destroy_process_group()
        return out
    except Exception as e:      # the headline number must not depend on this rehearsal
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass
        return dict(rccl_ranks=0, error='%s: %s' % (type(e).__name__, e))


def pmc_traffic(B, S):
    """HBM-side bytes per launch of the dominant kernel family from the newest committed rocprofv3 PMC passes (launch-weighted
    over its kernels) -- only if the kernel sources are still the ones that were profiled (fingerprint recorded with the pass)."""
    from face_vijnana_yolov3_amd.build import source_fingerprint
    for name in TRAFFIC_FILES:
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', name)))
        except (OSError, ValueError):
            continue
        parts = dominant_parts({k: v for k, v in tj.items() if isinstance(v, dict)})
        if not parts or B != PER_GPU_BATCH or S != IMAGE_SIZE:
            return None, None
        if tj.get('_source_fingerprint') != source_fingerprint():
            return None, 'profiles/%s was taken on other kernel sources (fingerprint %s, now %s): traffic not reported' % (
                name, tj.get('_source_fingerprint'), source_fingerprint())
        n = sum(v['launches_in_2_steps'] for v in parts.values())
        mb = sum((v['fetch_MB_per_launch'] + v['write_MB_per_launch']) * v['launches_in_2_steps'] for v in parts.values())
        return round(mb / n * 1e6), (
            'profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2 (gfx950 correction); '
            'fabric-side requests, Infinity-Cache hits included; launch-weighted over %s' % (name, ' + '.join(sorted(parts))))
    return None, None


def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as FRESH child processes through
    torch.distributed.run (one process per GPU, RCCL rendezvous on 127.0.0.1) BEFORE this process makes any GPU call,
    relay rank 0's JSON line on stdout (everything else goes to stderr) and return the launcher's exit code."""
    from face_vijnana_yolov3_amd.parallel import launch_ranks
    argv = [a for a in sys.argv[1:] if a != '--spawn']
    got = []

    def relay(line):
        if line.startswith('{"metric"'):
            sys.stdout.write(line); sys.stdout.flush(); got.append(1)
        else:
            sys.stderr.write(line); sys.stderr.flush()

    rc = launch_ranks(args.gpus, [os.path.abspath(__file__)] + argv, {'FV_BENCH_SELF_LAUNCHED': '1'}, relay)
    if rc == 0 and not got:
        print('bench.py: the ranks exited 0 without printing a result line', file=sys.stderr)
        rc = 1
    return rc


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and (args.gpus > 1 or args.spawn):
        sys.exit(self_launch(args))              # nothing above touched the GPU: the children own the devices
    # Rank 0's JSON line is the ONLY thing on stdout: C libraries write to fd 1 too (RCCL prints its version banner there at
    # the first communicator), so fd 1 is pointed at stderr for the life of the worker and the line goes to the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'FV_BENCH_DEVICE' in os.environ:   # rehearsal aid: several ranks on one GPU (not a measurement)
        local_rank = int(os.environ['FV_BENCH_DEVICE'])
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE=%d but --gpus %d' % (world, args.gpus))
    if args.rendezvous_only:
        # start-up rehearsal (tests/test_bench_launcher.py runs it with 2 ranks on CPU): rendezvous, one all-reduce, one line
        backend = os.environ.get('FV_DIST_BACKEND', 'nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
            t = torch.ones(1, device='cuda') * (rank + 1)
        else:
            dist.init_process_group(backend)
            t = torch.ones(1) * (rank + 1)
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            os.write(result_fd, (json.dumps({'metric': 'rendezvous-only', 'value': float(t.item()), 'n_gpus': world, 'backend': backend}) + '\n').encode())
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local_rank)
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.engine import Engine, train_flops_per_image
    from face_vijnana_yolov3_amd.parallel import DataParallelTrainer

    eng = Engine(local_rank)
    eng.init_synthetic(seed=7)                      # identical weights on every rank
    if args.no_overlap:
        eng.ctx.set_overlap(False)
    if args.no_tail_split:
        eng.ctx.set_tail_split(False)
    trainer = DataParallelTrainer(eng, world_size=world, rank=rank)  # inits RCCL when world > 1
    B, S = args.batch, args.image_size
    g = torch.Generator(device='cpu').manual_seed(1234 + rank)
    x = torch.rand((B, S, S, 3), generator=g).cuda()
    y = torch.from_numpy(data.synth_gt_batch(B, S, seed=1234 + rank)).cuda()

    def step():
        return trainer.train_on_batch(x, y, **HPS)

    for _ in range(args.warmup):
        step()
    for _ in range(64):                             # comm_mode 'auto' (N > 1): the choice is made before the timed region
        if not trainer.calibrating:
            break
        step()
    # per-step marks for the median: one event record per step on the launch stream (no host sync inside the region)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    trainer.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
    trainer.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = trainer.max_over_ranks(dt)
    loss_v = float(loss.item())
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = trainer.max_over_ranks(per_step[len(per_step) // 2] if len(per_step) % 2 else
                                       0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2]))

    # ---- N > 1: what the communication costs (every rank measures, rank 0 reports all of them)
    multi = None
    if world > 1:
        k = max(3, min(args.steps, 10))
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        allreduce_ms = trainer.allreduce_ms()        # one step's collectives back to back, nothing overlapping them
        torch.cuda.synchronize(); trainer.barrier()
        t1 = time.perf_counter()
        for _ in range(k):
            eng.train_on_batch(x, y, **HPS)          # the same step without any collective (replicas diverge in rounding only;
        torch.cuda.synchronize()                     # nothing is measured after this block that depends on the weights)
        compute_ms = (time.perf_counter() - t1) / k * 1e3
        # the same data-parallel step with the collectives BLOCKING on the compute stream (no overlap, no cross-stream traffic):
        # on one GPU that form is 2.5 ms per step cheaper than any overlapped one (parallel.DataParallelTrainer); which wins here?
        alt = DataParallelTrainer(eng, world_size=world, rank=rank, comm_mode='main' if trainer.comm_mode != 'main' else 'wg')
        for _ in range(3):
            alt.train_on_batch(x, y, **HPS)
        torch.cuda.synchronize(); trainer.barrier()
        t2 = time.perf_counter()
        for _ in range(k):
            alt.train_on_batch(x, y, **HPS)
        trainer.barrier(); torch.cuda.synchronize()
        alt_ms = trainer.max_over_ranks((time.perf_counter() - t2) / k * 1e3)
        mine = dict(rank=rank, allreduce_ms=round(allreduce_ms, 3), compute_only_ms=round(compute_ms, 3),
                    exposed_comm_ms=round(dt / args.steps * 1e3 - compute_ms, 3))
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        multi = dict(rccl_ranks=dist.get_world_size(), backend=dist.get_backend(), bucket_mib=trainer.bucket_bytes >> 20, comm_mode=trainer.comm_mode,
                     auto=trainer.auto_report, gradient_mb=round(eng.n_params * 4 / 1e6, 2), per_rank=allr,
                     alt_comm_mode=alt.comm_mode, alt_ms_per_step=round(alt_ms, 3),
                     alt_images_per_sec=round(world * B / (alt_ms * 1e-3), 2))
        for t in (eng.params, eng.state, eng.m, eng.v):
            dist.broadcast(t, 0)
        trainer.barrier()

    if world == 1 and not args.no_rccl_rehearsal:
        multi = rccl_world1_rehearsal(eng, x, y)     # right after the timed region: same clocks, same allocator state

    # instrumented steps for the roofline of the dominant kernel (HIP events on the launch stream).
    # They run with fv_set_option("overlap", 0): under the backward overlap two MFMA kernels time-share the
    # chip and a launch's elapsed time is no longer that kernel's own rate (the timed region above
    # keeps the overlap; `roofline_overlapped` repeats the measurement with it on).  EVERY rank runs
    # these steps (they contain the gradient collectives); only rank 0 records events.
    prof, prof_ov = {}, {}
    if args.profile_steps > 0:
        for on, store in ((False, prof), (True, prof_ov)):
            if args.no_overlap and on:
                continue
            eng.ctx.set_overlap(on)
            step(); torch.cuda.synchronize()
            if rank == 0:
                eng.ctx.profile(True)
            for _ in range(args.profile_steps):
                step()
            if rank == 0:
                store.update(eng.ctx.profile_collect())
                eng.ctx.profile(False)
        eng.ctx.set_overlap(not args.no_overlap)
        trainer.barrier()
    out = None
    if rank == 0:
        detect = None if args.no_detect else detect_bench(eng, x)
        if detect is not None and world == 1 and not args.no_loader:
            detect['test_loop'] = test_loop_bench(local_rank, S)
            detect['test_loop_three_scale_head'] = test_loop_bench(local_rank, S, n_img=256, head='three_scale')
        # The secondary sections run IN this process again.  Round 3 moved them into fresh child processes because steps measured
        # late in a process had run 13-30 % slow (three-scale 31.6 -> 41 ms, base 53 -> 60 ms); in round 4 that slowdown reproduced
        # with none of: the original reproducer, these sections in any order, contexts created late, pinned-memory churn
        # (profiles/r04_slowdown_probe.txt: every step within 1 % of its fresh-process time).  `process_history_check` below
        # re-measures the headline step at the very end of this process, so a recurrence shows up in the line itself.
        loader = None
        if world == 1 and not args.no_loader:
            loader = loader_bench(eng, trainer, B, S, args.loader_steps)
        three = None
        if world == 1 and not args.no_three_scale and not args.no_detect:
            three = three_scale_bench(local_rank, S, B=B, steps=5)
            three['batch16'] = {k: v for k, v in three_scale_bench(local_rank, S, B=16, steps=5).items()
                                if k in ('value', 'ms_per_step', 'step_tflops', 'frac_of_fp32_mfma_peak')}
        config5 = None
        if world == 1 and not args.no_three_scale and not args.no_detect and (B, S) == (PER_GPU_BATCH, IMAGE_SIZE):
            config5 = config5_bench(local_rank)
        history = None
        if world == 1 and args.profile_steps > 0:
            torch.cuda.synchronize()
            late = []
            for _ in range(3):
                step()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
            ev[0].record()
            for i in range(10):
                step(); ev[i + 1].record()
            torch.cuda.synchronize()
            late = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10))
            late_ms = 0.5 * (late[4] + late[5])
            history = dict(fresh_median_ms_per_step=round(median_ms, 3), late_median_ms_per_step=round(late_ms, 3),
                           ratio=round(late_ms / median_ms, 4),
                           note='the headline step re-timed (10 steps, median) after every other section of this process has run')
        dom_name, dom = dominant(prof)
        roofline = None
        traffic, traffic_source = pmc_traffic(B, S)
        if dom and dom['ms'] > 0:
            ach = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
            roofline = dict(bound='mfma', achieved=round(ach, 2), peak=FP32_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
                            frac=round(ach / FP32_MFMA_PEAK_TFLOPS, 4), traffic=traffic, traffic_source=traffic_source,
                            kernel=dom_name,
                            algorithmic_bytes_per_launch=round(dom['bytes'] / dom['launches']),
                            mode='exclusive: instrumented steps run with fv_set_option("overlap", 0)',
                            launches_per_step=dom['launches'] // max(args.profile_steps, 1),
                            avg_launch_ms=round(dom['ms'] / dom['launches'], 4),
                            gflop_per_launch=round(dom['flops'] / dom['launches'] / 1e9, 3),
                            parts={k: dict(launches_per_step=v['launches'] // max(args.profile_steps, 1),
                                           avg_launch_ms=round(v['ms'] / v['launches'], 4),
                                           gflop_per_launch=round(v['flops'] / v['launches'] / 1e9, 3),
                                           achieved=round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2))
                                   for k, v in dominant_parts(prof).items()})
        kernels = {k: dict(launches=v['launches'] // max(args.profile_steps, 1),
                           ms_per_step=round(v['ms'] / max(args.profile_steps, 1), 3),
                           tflops=round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2) if v['flops'] and v['ms'] else None,
                           gbps=round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) if v['ms'] else None)
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])}
        train_flops = train_flops_per_image(S)       # fwd + wgrad(all) + dgrad(all but conv_0)
        ips = world * B * args.steps / dt
        out = {
            'metric': 'training images/sec (%dx%d bs=%d per GPU)' % (S, S, B), 'value': round(ips, 2), 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'median_ms_per_step': round(median_ms, 3), 'images_per_sec_at_median': round(world * B / (median_ms * 1e-3), 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'FaceDetector mode=train image_size=%d batch_size=%d per GPU, synthetic UCCS-shaped '
                                   'batch, random-init Darknet-53 base + %dx%dx6 head, MSE, Adam lr 1e-4 b1=b2=0.99' % (S, B, S // 32, S // 32),
                       'global_batch': world * B, 'parallelism': 'dp%d' % world},
            'loss': loss_v,
            'step_tflops_per_gpu': round(train_flops * B * args.steps / dt / 1e12, 2),
            'step_frac_of_fp32_mfma_peak': round(train_flops * B * args.steps / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            'roofline': roofline,
            'roofline_overlapped': (lambda d: None if not d or not d['ms'] else dict(
                achieved=round(d['flops'] / (d['ms'] * 1e-3) / 1e12, 2), avg_launch_ms=round(d['ms'] / d['launches'], 4)))(dominant(prof_ov)[1]),
            # the HBM-bound companions (algorithmic bytes / HIP-event time, against the 8 TB/s HBM3E peak)
            'roofline_hbm': {k: dict(bound='hbm', achieved=v['gbps'], peak=HBM_PEAK_GBPS, unit='GB/s',
                                     frac=round(v['gbps'] / HBM_PEAK_GBPS, 4), ms_per_step=v['ms_per_step'])
                             for k, v in kernels.items()
                             if k in ('bn_act_stats_kernel', 'bn_bwd_apply_slots_kernel', 'bn_bwd_reduce_kernel', 'adam_kernel') and v['gbps']},
            'detect': detect,
            'loader_inclusive': loader,
            'three_scale_train': three,
            'config5_608_bs16': config5,
            'process_history_check': history,
            'multi_gpu': multi,
            'kernels': kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.cpu_batch, S)
        else:
            out['cpu_baseline'] = None
    trainer.shutdown()
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + '\n').encode())


if __name__ == '__main__':
    main()
