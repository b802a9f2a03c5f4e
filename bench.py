#!/usr/bin/env python3
"""Headline benchmark: FaceDetector training images/sec at 416x416, 40 images per GPU
(BASELINE.json configs[1]; N GPUs -> global batch 40*N, weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = forward (training-mode BN) + MSE + backward + (N>1: RCCL all-reduce of the 40.64 M fp32
gradients, bucketed and overlapped with backward) + Keras-formula Adam, on synthetic inputs that
are already resident in HBM.  Rank 0 prints ONE JSON line.

`roofline`: the dominant kernel is the 128x128-tile fp32-MFMA implicit-GEMM conv
(conv_kernel<128,2,2,false>: forward and data-gradient of every layer with >= 128 output
channels).  achieved = algorithmic FLOPs of its launches / their HIP-event-timed duration, taken in
instrumented steps right after the timed region (the timed steps themselves run un-instrumented).
peak = 157.3 TFLOP/s, the dense fp32 MFMA rate of MI355X (MI355X_MICROARCH.md).
`cpu_baseline`: the torch-CPU oracle restatement of the same step (kind "port"; the Keras/TF
reference cannot run here) on a bounded sample, timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IMAGE_SIZE = 416
PER_GPU_BATCH = 40
FP32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (about 6.3 TB/s is what a float4 copy reaches)
DOMINANT = 'conv_kernel<128,2,2,false>'
HPS = dict(lr=1e-4, beta_1=0.99, beta_2=0.99, decay=0.0)  # reference face_vijnana_yolov3.json:12-15


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=PER_GPU_BATCH, help='per-GPU batch (default 40 = BASELINE config)')
    ap.add_argument('--image-size', type=int, default=IMAGE_SIZE)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=1)
    ap.add_argument('--profile-steps', type=int, default=2)
    ap.add_argument('--no-overlap', action='store_true', help='serialise wgrad on the main stream (A/B aid)')
    ap.add_argument('--no-tail-split', action='store_true', help='conv launches without the tail split (A/B aid)')
    ap.add_argument('--no-detect', action='store_true', help='skip the detect-path measurement (PMC passes)')
    return ap.parse_args()


def cpu_baseline(batch, image_size):
    """Torch-CPU oracle train step (fwd+bwd+Adam) on a bounded sample: `batch` images."""
    import torch
    from oracle import net_oracle as no
    torch.manual_seed(0)
    p, st = no.init_params(7, torch.float32)
    x = torch.rand((batch, image_size, image_size, 3))
    g = image_size // 32
    yt = torch.rand((batch, g, g, 6))
    m = torch.zeros_like(p); v = torch.zeros_like(p)
    t0 = time.time()
    loss, grad, st = no.train_step_grads(p, st, x, yt)
    p, m, v = no.keras_adam(p, grad, m, v, 0, **{k: HPS[k] for k in ('lr', 'beta_1', 'beta_2')})
    dt = time.time() - t0
    return dict(value=batch / dt, unit='images/sec', cores=torch.get_num_threads(), kind='port',
                sample='1 train step (fwd+bwd+Adam) of the torch-CPU oracle at batch %d, %dx%d, fp32; %.1f s'
                       % (batch, image_size, image_size, dt))


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
    import time as _t
    t0 = _t.perf_counter()
    for _ in range(20):
        to_boundboxes(one(x1), 0)           # end to end incl. D2H + BoundBox objects, as detect() returns
    e2e = (_t.perf_counter() - t0) / 20 * 1e3
    rng = np.random.default_rng(99)
    head = np.zeros((10000, 13, 13, 6), np.float32)
    head[..., 0] = rng.normal(0, 2, (10000, 13, 13)); head[..., 5] = rng.normal(0, 2, (10000, 13, 13))
    head[..., 1:3] = rng.uniform(0, 1, (10000, 13, 13, 2)); head[..., 3:5] = rng.uniform(0, 0.3, (10000, 13, 13, 2))
    hd = torch.from_numpy(head).cuda()
    tpp = timed(lambda: decode_nms(eng.ctx, hd, 416, 0.5, 0.5, 60), 10)
    return dict(unit='ms/img', batch1_device=round(t1, 4), batch1_end_to_end=round(e2e, 4),
                batch40_device=round(t40 / x40.shape[0], 4), postproc_10k_frames=round(tpp / 10000, 6),
                postproc_10k_total_ms=round(tpp, 3))


def main():
    args = parse()
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'FV_BENCH_DEVICE' in os.environ:   # rehearsal aid: several ranks on one GPU (not a measurement)
        local_rank = int(os.environ['FV_BENCH_DEVICE'])
    if world != args.gpus and world > 1:
        raise SystemExit('WORLD_SIZE=%d but --gpus %d' % (world, args.gpus))
    if args.gpus > 1 and world == 1:
        raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d '
                         '--master-addr 127.0.0.1 --master-port 29500 bench.py --gpus %d ...' % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.engine import Engine
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
    trainer.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    trainer.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = trainer.max_over_ranks(dt)
    loss_v = float(loss.item())

    # instrumented steps for the roofline of the dominant kernel (HIP events on the launch stream).
    # They run with fv_set_overlap(0): under the backward overlap two MFMA kernels time-share the
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
        dom = prof.get(DOMINANT)
        roofline = None
        traffic = None   # HBM-side bytes per launch from committed rocprofv3 PMC passes (cannot be read live)
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')))
            tk = tj.get('conv_kernel<128, 2, 2, false>')
            if tk and B == PER_GPU_BATCH and S == IMAGE_SIZE:
                traffic = round((tk['fetch_MB_per_launch'] + tk['write_MB_per_launch']) * 1e6)   # bytes per launch
        except (OSError, ValueError, KeyError):
            pass
        traffic_source = None if traffic is None else (
            'profiles/r01_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2 '
            '(gfx950 correction); fabric-side requests, Infinity-Cache hits included')
        if dom and dom['ms'] > 0:
            ach = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
            roofline = dict(bound='mfma', achieved=round(ach, 2), peak=FP32_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
                            frac=round(ach / FP32_MFMA_PEAK_TFLOPS, 4), traffic=traffic, traffic_source=traffic_source,
                            kernel=DOMINANT,
                            algorithmic_bytes_per_launch=round(dom['bytes'] / dom['launches']),
                            mode='exclusive: instrumented steps run with fv_set_overlap(0)',
                            launches_per_step=dom['launches'] // max(args.profile_steps, 1),
                            avg_launch_ms=round(dom['ms'] / dom['launches'], 4),
                            gflop_per_launch=round(dom['flops'] / dom['launches'] / 1e9, 3))
        kernels = {k: dict(launches=v['launches'] // max(args.profile_steps, 1),
                           ms_per_step=round(v['ms'] / max(args.profile_steps, 1), 3),
                           tflops=round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2) if v['flops'] and v['ms'] else None,
                           gbps=round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) if v['ms'] else None)
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])}
        from oracle import net_oracle as no
        fwd = no.fwd_flops_per_image(S)
        conv0 = 2 * S * S * 27 * 32
        train_flops = 3 * fwd - conv0  # fwd + wgrad(all) + dgrad(all but conv_0)
        ips = world * B * args.steps / dt
        out = {
            'metric': 'training images/sec (416x416 bs=40 per GPU)', 'value': round(ips, 2), 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'FaceDetector mode=train image_size=%d batch_size=%d per GPU, synthetic UCCS-shaped '
                                   'batch, random-init Darknet-53 base + 13x13x6 head, MSE, Adam lr 1e-4 b1=b2=0.99' % (S, B),
                       'global_batch': world * B, 'parallelism': 'dp%d' % world},
            'loss': loss_v,
            'step_tflops_per_gpu': round(train_flops * B * args.steps / dt / 1e12, 2),
            'step_frac_of_fp32_mfma_peak': round(train_flops * B * args.steps / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            'roofline': roofline,
            'roofline_overlapped': (lambda d: None if not d or not d['ms'] else dict(
                achieved=round(d['flops'] / (d['ms'] * 1e-3) / 1e12, 2), avg_launch_ms=round(d['ms'] / d['launches'], 4)))(prof_ov.get(DOMINANT)),
            # the HBM-bound companions (algorithmic bytes / HIP-event time, against the 8 TB/s HBM3E peak)
            'roofline_hbm': {k: dict(bound='hbm', achieved=v['gbps'], peak=HBM_PEAK_GBPS, unit='GB/s',
                                     frac=round(v['gbps'] / HBM_PEAK_GBPS, 4), ms_per_step=v['ms_per_step'])
                             for k, v in kernels.items()
                             if k in ('bn_act_stats_kernel', 'bn_bwd_apply_slots_kernel', 'bn_bwd_reduce_kernel', 'adam_kernel') and v['gbps']},
            'detect': detect,
            'kernels': kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.cpu_batch, S)
        else:
            out['cpu_baseline'] = None
    trainer.shutdown()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
