"""Which kernels stretch when a training step runs late in a process?  

Reproducer = the sequence of tools/stream_env_probe.py (three-scale step 31.6 -> 41 ms, base step 53 -> 60 ms).  This probe takes,
in the FAST state (fresh process) and in the SLOW state (after the perturbing sequence), for the three-scale step and the base
step: the step time (overlapped schedule), the per-kernel HIP-event table in the serial schedule (fv_profile_*), a device copy
(HBM yardstick) and one MFMA-bound conv launch (clock yardstick).  Then it tries to get back to the fast state: free the other
models, empty torch's cache, re-allocate the workspace, allocate the workspace with one raw hipMalloc.

    python tools/slowdown_probe.py [--stage stream|rccl|ctx|testloop|base|all]
"""
import argparse
import ctypes
import os
import sys
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--stages', default='stream,rccl,ctx,testloop,base')
    ap.add_argument('--each', action='store_true', help='measure after every stage, not only at the end')
    args = ap.parse_args()
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from face_vijnana_yolov3_amd import data, ops
    from face_vijnana_yolov3_amd._lib import lib
    from face_vijnana_yolov3_amd.engine import Engine
    from face_vijnana_yolov3_amd.yolov3 import Yolov3
    import bench

    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))

    def mem_info():
        fr, tot = ctypes.c_size_t(0), ctypes.c_size_t(0)
        hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot))
        return fr.value / 2**30, tot.value / 2**30

    m = Yolov3(0, out_channels=255); m.init_synthetic(3)
    g = torch.Generator().manual_seed(4)
    x16 = torch.rand((16, 416, 416, 3), generator=g).cuda()
    tg = [torch.rand((16, 416 // d, 416 // d, 255), generator=g).cuda() for d in (32, 16, 8)]
    e1 = Engine(0); e1.init_synthetic(7)
    x40 = torch.rand((40, 416, 416, 3)).cuda(); y40 = torch.from_numpy(data.synth_gt_batch(40, 416, seed=1)).cuda()
    cp_src = torch.empty(256 << 20, dtype=torch.float32, device='cuda').normal_()     # 1 GiB
    cp_dst = torch.empty_like(cp_src)

    def timed(fn, n):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1_ = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1_.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1_) / n

    def table(ctx, fn, n=2):
        ctx.set_overlap(False)
        fn(); torch.cuda.synchronize()
        ctx.profile(True)
        for _ in range(n):
            fn()
        t = ctx.profile_collect()
        ctx.profile(False)
        ctx.set_overlap(True)
        return {k: v['ms'] / n for k, v in t.items()}

    tables = {}

    def measure(label):
        three = timed(lambda: m.train_on_batch(x16, tg, 1e-4, 0.9, 0.99), 5)
        base = timed(lambda: e1.train_on_batch(x40, y40, **bench.HPS), 5)
        cp = timed(lambda: cp_dst.copy_(cp_src), 10)
        fr, tot = mem_info()
        print('%-46s three-scale %.2f ms  base %.2f ms  copy %.0f GB/s  torch reserved %.1f GiB  device free %.1f / %.1f GiB' % (
            label, three, base, 2 * cp_src.numel() * 4 / cp / 1e6, torch.cuda.memory_reserved() / 2**30, fr, tot), flush=True)
        tables[label] = (table(m.ctx, lambda: m.train_on_batch(x16, tg, 1e-4, 0.9, 0.99)),
                         table(e1.ctx, lambda: e1.train_on_batch(x40, y40, **bench.HPS)))
        return three, base

    def compare(a, b):
        for which, name in ((0, 'three-scale step'), (1, 'base step')):
            ta, tb = tables[a][which], tables[b][which]
            print('--- %s, serial schedule, ms per step: %s -> %s' % (name, a, b))
            tot_a = tot_b = 0.0
            for k in sorted(ta, key=lambda k: -ta[k]):
                if k in tb:
                    tot_a += ta[k]; tot_b += tb[k]
                    if ta[k] > 0.05:
                        print('    %-34s %8.3f -> %8.3f   x%.3f' % (k, ta[k], tb[k], tb[k] / ta[k]))
            print('    %-34s %8.3f -> %8.3f   x%.3f' % ('sum', tot_a, tot_b, tot_b / tot_a), flush=True)

    def late_models(label):
        """Models whose CONTEXTS (side stream, events) are created now: is it the creation order that decides?"""
        m2 = Yolov3(0, out_channels=255); m2.init_synthetic(3)
        e3 = Engine(0); e3.init_synthetic(7)
        three = timed(lambda: m2.train_on_batch(x16, tg, 1e-4, 0.9, 0.99), 5)
        base = timed(lambda: e3.train_on_batch(x40, y40, **bench.HPS), 5)
        e3.ctx.set_overlap(False); m2.ctx.set_overlap(False)
        three_s = timed(lambda: m2.train_on_batch(x16, tg, 1e-4, 0.9, 0.99), 3)
        base_s = timed(lambda: e3.train_on_batch(x40, y40, **bench.HPS), 3)
        print('%-46s NEW contexts: three-scale %.2f ms  base %.2f ms   (weight-gradients on the compute stream: %.2f / %.2f)' % (
            label, three, base, three_s, base_s), flush=True)
        del m2, e3
        torch.cuda.empty_cache()

    measure('fresh')
    late_models('fresh')
    stages = args.stages.split(',')
    keep = []
    for st in stages:
        if st == 'stream':
            s = torch.cuda.Stream(); s.synchronize(); keep.append(s)
        elif st == 'rccl':
            dist.init_process_group('nccl', store=dist.HashStore(), rank=0, world_size=1, device_id=torch.device('cuda', 0))
            t = torch.ones(1024, device='cuda'); dist.all_reduce(t); torch.cuda.synchronize()
            dist.destroy_process_group()
        elif st == 'ctx':
            engs = [Engine(0) for _ in range(3)]
            for e in engs:
                e.init_synthetic(1); e.predict_device(x16[:1])
            keep.append(engs)
        elif st == 'testloop':
            bench.test_loop_bench(0, 416, n_img=16)
        elif st == 'base':
            e2 = Engine(0); e2.init_synthetic(7)
            for _ in range(3):
                e2.train_on_batch(x40, y40, **bench.HPS)
            torch.cuda.synchronize()
            keep.append(e2)
        elif st == 'rehearsal':
            print(bench.rccl_world1_rehearsal(e1, x40, y40))
        elif st == 'pinchurn':        # what the round-3 loaders did before PinnedRing: a fresh page-locked staging buffer per batch
            import numpy as np
            rng = np.random.default_rng(0)
            held = []
            for i in range(60):
                t = torch.empty(int(rng.integers(35, 100)) << 20, dtype=torch.uint8).pin_memory()
                t.cuda(non_blocking=True)
                held.append(t)
                if len(held) > 3:
                    held.pop(0)
            torch.cuda.synchronize()
            keep.append(held)
        elif st == 'loader':          # bench.loader_bench in this process (BatchFeeder: 16 host threads, pinned ring, staging stream)
            from face_vijnana_yolov3_amd.parallel import DataParallelTrainer
            print(bench.loader_bench(e1, DataParallelTrainer(e1, world_size=1, rank=0), 40, 416, 8))
        elif st == 'detect':
            print(bench.detect_bench(e1, x40))
        elif st == 'threescale':      # bench.three_scale_bench in this process (a second three-scale model + workspace)
            print(bench.three_scale_bench(0, 416, steps=3))
        if args.each:
            measure('after ' + st)
            late_models('after ' + st)
    if not args.each:
        measure('after ' + '+'.join(stages))
    last = list(tables)[-1]
    compare('fresh', last)
    late_models(last)
    # ---- back to the fast state?
    del keep[:]
    import gc; gc.collect()
    torch.cuda.synchronize()
    measure('others deleted (cache kept)')
    torch.cuda.empty_cache()
    measure('torch cache emptied')
    m._tws = {}; e1._ws = {}
    torch.cuda.empty_cache()
    measure('own workspaces re-allocated')
    compare('fresh', 'own workspaces re-allocated')


if __name__ == '__main__':
    main()
