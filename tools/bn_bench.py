"""HBM rate of the two BN passes of the training step (bn_act_stats, bn_bwd_apply_slots) over the activation shapes of
Darknet-53 at batch 40, 416x416, through the C ABI; also a device copy of the same bytes as the yardstick of the box."""
import os, sys

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from face_vijnana_yolov3_amd import ops
    from face_vijnana_yolov3_amd._lib import Context
    ctx = Context(0)
    shapes = [(40 * 416 * 416, 32, 1), (40 * 208 * 208, 64, 2), (40 * 208 * 208, 32, 1), (40 * 104 * 104, 128, 3), (40 * 104 * 104, 64, 2),
              (40 * 52 * 52, 256, 9), (40 * 52 * 52, 128, 8), (40 * 26 * 26, 512, 9), (40 * 26 * 26, 256, 8), (40 * 13 * 13, 1024, 5), (40 * 13 * 13, 512, 4)]
    def timeit(fn, reps=10):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    tot = [0.0, 0.0, 0.0]; byt = [0.0, 0.0, 0.0]
    for rows, C, cnt in shapes:
        z = torch.randn((rows, C), device='cuda'); g = torch.randn((rows, C), device='cuda')
        gamma = torch.rand(C, device='cuda') + 0.5; beta = torch.randn(C, device='cuda')
        sl = ops.stat_slots(C, 'cuda'); sl[0, 0] = z.double().sum(0); sl[0, 1] = (z.double() ** 2).sum(0)
        out, mean, invstd, scale, shift = ops.bn_act_slots(ctx, z, sl, gamma, beta)
        t_a = timeit(lambda: ops.bn_act_slots(ctx, z, sl, gamma, beta))
        bs = ops.stat_slots(C, 'cuda')
        t_b = timeit(lambda: ops.bn_bwd_slots(ctx, g, z, scale, shift, mean, invstd, bs, True))
        dst = torch.empty_like(z)
        t_c = timeit(lambda: dst.copy_(z))
        n = rows * C * 4
        print('%9d x %4d (x%d): bn_act %.3f ms %5.0f GB/s | bn_bwd %.3f ms %5.0f GB/s | copy %.3f ms %5.0f GB/s' %
              (rows, C, cnt, t_a, 2 * n / t_a / 1e6, t_b, 3 * n / t_b / 1e6, t_c, 2 * n / t_c / 1e6), flush=True)
        for k, (t, m) in enumerate(((t_a, 2), (t_b, 3), (t_c, 2))):
            tot[k] += t * cnt; byt[k] += m * n * cnt
    print('network: bn_act %.3f ms %.0f GB/s | bn_bwd %.3f ms %.0f GB/s | copy %.3f ms %.0f GB/s' %
          (tot[0], byt[0] / tot[0] / 1e6, tot[1], byt[1] / tot[1] / 1e6, tot[2], byt[2] / tot[2] / 1e6))


if __name__ == '__main__':
    main()
