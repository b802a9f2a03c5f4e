#!/usr/bin/env python3
"""Dev tool: run the same BASELINE-size train step repeatedly and report per-layer run-to-run
differences of the gradients (overlap on / off).  Anything beyond float-atomic noise in dW
(~1e-6 relative) would indicate a race."""
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
    for ov in (False, True):
        eng.ctx.set_overlap(ov)
        runs = []
        for r in range(3):
            eng.set_params(p0, s0)
            l = eng.forward_backward(x, y).item()
            torch.cuda.synchronize()
            runs.append((l, eng.grads.clone(), eng.state.clone()))
        worst = []
        for d in eng.layers:
            n = d['cout'] * d['ksize'] ** 2 * d['cin']
            a = runs[0][1][d['w_off']:d['w_off'] + n]
            rel = max(((runs[k][1][d['w_off']:d['w_off'] + n] - a).abs().max() / a.abs().max()).item() for k in (1, 2))
            gb = 0.0
            if d['has_bn']:
                c = d['cout']
                a2 = runs[0][1][d['gamma_off']:d['gamma_off'] + 2 * c]
                gb = max(((runs[k][1][d['gamma_off']:d['gamma_off'] + 2 * c] - a2).abs().max() / a2.abs().max()).item() for k in (1, 2))
            worst.append((rel, gb, d['darknet_index']))
        print('overlap', ov, 'losses', [r[0] for r in runs], 'state equal', torch.equal(runs[0][2], runs[1][2]))
        print('  max rel dW diff', max(w[0] for w in worst), 'max rel dgamma/dbeta diff', max(w[1] for w in worst))
        print('  worst layers', sorted(worst, reverse=True)[:5])


if __name__ == '__main__':
    main()
