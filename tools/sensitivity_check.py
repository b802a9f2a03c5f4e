#!/usr/bin/env python3
"""Dev tool: how strongly does the random-init network amplify an fp32-rounding-sized input
perturbation into the gradients?  (Explains the batch-permutation tolerance in
tests/test_fullsize_gpu.py: the problem is ill-conditioned, the kernels are deterministic.)"""
import os, sys

def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from face_vijnana_yolov3_amd import data
    from face_vijnana_yolov3_amd.engine import Engine

    B, S = 40, 416
    eng = Engine(0); eng.init_synthetic(7)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand((B, S, S, 3), generator=g).cuda()
    y = torch.from_numpy(data.synth_gt_batch(B, S, seed=1234)).cuda()
    p0, s0 = eng.params.clone(), eng.state.clone()
    eng.forward_backward(x, y); torch.cuda.synchronize(); g0 = eng.grads.clone()
    for eps in (1e-7, 1e-6):
        eng.set_params(p0, s0)
        xp = x * (1 + eps * torch.randn(x.shape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1)))
        eng.forward_backward(xp, y); torch.cuda.synchronize(); g1 = eng.grads.clone()
        out = []
        for li, d in enumerate(eng.layers):
            n = d['cout'] * d['ksize'] ** 2 * d['cin']
            a, b = g0[d['w_off']:d['w_off'] + n], g1[d['w_off']:d['w_off'] + n]
            out.append(((a - b).abs().max() / a.abs().max()).item())
        print('input perturbation %.0e -> rel dW change: conv_0 %.2e, layer 26 %.2e, layer 47 %.2e, head %.2e, max %.2e'
              % (eps, out[0], out[26], out[47], out[-1], max(out)))


if __name__ == '__main__':
    main()
