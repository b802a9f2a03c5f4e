"""Diagnostic for VERDICT r2 weak #1: per-layer forward / gradient errors of the three-scale and the base
train step at tiny grids (2x2 cells at the coarsest scale), with every schedule switch toggled."""
import sys
import torch
sys.path.insert(0, '.')
from face_vijnana_yolov3_amd.yolov3 import Yolov3
from face_vijnana_yolov3_amd._lib import lib
import ctypes
from oracle import net_oracle as no
sys.path.insert(0, 'tests')


def setup(out_ch, B, S, seed):
    p64, s64 = no.yolov3_init(seed, out_ch, torch.float64)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.rand((B, S, S, 3), dtype=torch.float64, generator=g)
    tg = []
    for d in (32, 16, 8):
        t = torch.rand((B, S // d, S // d, out_ch), dtype=torch.float64, generator=g)
        t4 = t.view(B, S // d, S // d, 3, out_ch // 3)
        t4[..., 4] = (t4[..., 4] > 0.8).double(); t4[..., 5:] = (t4[..., 5:] > 0.7).double()
        tg.append(t)
    return p64, s64, x, tg


def wtensor(model, B, S, l, code):
    ws = model._train_ws(B, S)
    off, cnt = ctypes.c_size_t(0), ctypes.c_int64(0)
    assert lib().fv_yolov3_train_workspace_tensor(B, S, model.out_channels, l, code, ctypes.byref(off), ctypes.byref(cnt)) == 0
    return ws[off.value:off.value + 4 * cnt.value].view(torch.float32)


def run(out_ch, B, S, seed=21, verbose=True, toggles=None):
    model = Yolov3(0, out_channels=out_ch)
    for k, v in (toggles or {}).items():
        getattr(model.ctx, 'set_' + k)(v)
    p64, s64, x, tg = setup(out_ch, B, S, seed)
    model.set_params(p64.float(), s64.float())
    loss = model.forward_backward(x.float(), [t.float() for t in tg])
    torch.cuda.synchronize()
    pos = [m.cpu() for m in model.leaky_slopes_taken(B, S)]
    cap = {}
    p = p64.clone().requires_grad_(True)
    ns = s64.clone()
    outs = no.yolov3_forward(p, s64, x, out_ch, training=True, positive=pos, new_state=ns, capture=cap)
    nclass = out_ch // 3 - 5
    l64 = sum(no.yolo_scale_loss(o, y, nclass) for o, y in zip(outs, tg))
    (g64,) = torch.autograd.grad(l64, p)
    l32, g32, _ = no.yolov3_train_step_grads(p64.float(), s64.float(), x.float(), [t.float() for t in tg], out_ch, positive=pos)
    ents, n, _ = no.yolov3_layout(out_ch)
    g = model.grads.cpu().double()
    worst = 0.0
    if verbose:
        print('loss dev %.8f  f64 %.8f  f32 %.8f' % (loss.item(), l64.item(), l32.item()))
    li = 0
    for l, e in enumerate(ents):
        k, cin, cout = e['k'], e['cin'], e['cout']
        line = '%-9s k%d %4d->%4d ' % (e['name'], k, cin, cout)
        if e['has_bn']:
            z = wtensor(model, B, S, l, 0).cpu().double()
            mean = wtensor(model, B, S, l, 2).cpu().double(); invstd = wtensor(model, B, S, l, 3).cpu().double()
            z64, m64, v64 = cap[e['name']]
            zr = (z - z64.reshape(-1)).norm().item() / max(z64.norm().item(), 1e-30)
            mr = (mean - m64).abs().max().item()
            i64 = 1.0 / torch.sqrt(v64 + 1e-3)
            ir = ((invstd - i64).abs() / i64).max().item()
            line += 'rows %5d z %.1e mean %.1e invstd %.1e | ' % (z64.numel() // cout, zr, mr, ir)
        parts = [('dW', slice(e['w_off'], e['w_off'] + cout * k * k * cin))]
        parts += [('dg', slice(e['gamma_off'], e['gamma_off'] + cout)), ('db', slice(e['beta_off'], e['beta_off'] + cout))] if e['has_bn'] \
            else [('dbias', slice(e['bias_off'], e['bias_off'] + cout))]
        for nm, sl in parts:
            n64 = g64[sl].norm().item()
            rel = (g[sl] - g64[sl]).norm().item() / max(n64, 1e-30)
            rel32 = (g32[sl].double() - g64[sl]).norm().item() / max(n64, 1e-30)
            line += '%s %.1e(%.1e)%s ' % (nm, rel, rel32, '!' if rel > max(6 * rel32, 4e-5) else '')
            worst = max(worst, rel)
        if verbose:
            print(line)
    print('== out_ch=%d B=%d S=%d toggles=%s worst %.3e' % (out_ch, B, S, toggles, worst), flush=True)
    del model
    return worst


def repeat_check(out_ch, B, S, n=6, seed=21):
    """The same step n times in one process: gradient tensors may differ run to run only by the float-atomic order (1e-6)."""
    model = Yolov3(0, out_channels=out_ch)
    p64, s64, x, tg = setup(out_ch, B, S, seed)
    model.set_params(p64.float(), s64.float())
    ref = None
    for i in range(n):
        model.forward_backward(x.float(), [t.float() for t in tg])
        torch.cuda.synchronize()
        g = model.grads.clone()
        if ref is None:
            ref = g
        else:
            ents, _, _ = no.yolov3_layout(out_ch)
            w = 0.0
            for e in ents:
                sl = slice(e['w_off'], e['w_off'] + e['cout'] * e['k'] ** 2 * e['cin'])
                w = max(w, ((g[sl] - ref[sl]).norm() / ref[sl].norm()).item())
            print('repeat %d: worst per-tensor rel diff vs run 0: %.2e' % (i, w), flush=True)


if __name__ == '__main__':
    torch.set_num_threads(16)
    repeat_check(27, 2, 64)
    run(27, 1, 64)
