"""BASELINE config 5 at its per-GPU size (image_size 608, batch 16: grid 19, 5 776 rows in the 19x19 layers = 46 M tiles whose
128-wide launches take the tail-split plans; 92 416 rows at 76x76) where the CPU oracle cannot run in seconds: the
size-independent properties tests/test_fullsize_gpu.py checks at 40 x 416^2, on the plans this size really runs.
(One image at 608 is compared with the oracle in tests/test_net_gpu.py::test_config5_608_grid19.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu

B, S = 16, 608


@pytest.fixture(scope='module')
def eng():
    from face_vijnana_yolov3_amd.engine import Engine
    e = Engine(0)
    e.init_synthetic(seed=7)
    return e


@pytest.fixture(scope='module')
def batch():
    from face_vijnana_yolov3_amd import data
    g = torch.Generator().manual_seed(608)
    x = torch.rand((B, S, S, 3), generator=g).cuda()
    y = torch.from_numpy(data.synth_gt_batch(B, S, seed=608)).cuda()
    return x, y


def test_608_inference_is_per_image_and_deterministic(eng, batch):
    """Inference-mode BN is per sample: the batch in two uneven parts (other tile counts, other tail plans, the K-split path for
    the small part's 19x19 layers) reproduces the full-batch result to fp32 rounding; repeated calls are bit-identical; the
    tail split on / off changes only the summation order of the tail tiles."""
    x, _ = batch
    y = eng.predict_device(x).clone()
    assert torch.equal(y, eng.predict_device(x))
    assert tuple(y.shape) == (B, 19, 19, 6) and bool(torch.isfinite(y).all())
    ys = torch.cat([eng.predict_device(x[:5].contiguous()).clone(), eng.predict_device(x[5:].contiguous()).clone()])
    assert (ys - y).abs().max().item() <= 2e-5 * y.abs().max().item()
    eng.ctx.set_tail_split(False)
    try:
        yn = eng.predict_device(x).clone()
    finally:
        eng.ctx.set_tail_split(True)
    assert (yn - y).abs().max().item() <= 2e-5 * y.abs().max().item()


def test_608_train_step_batch_permutation_invariance(eng, batch):
    """Batch statistics, the loss and every gradient are symmetric in the batch order (bounds as at 416: the head tight, the base
    layers at the measured conditioning of the randomly initialised network)."""
    x, y = batch
    p0, s0 = eng.params.clone(), eng.state.clone()
    eng.grads = eng.m = eng.v = None
    l1 = eng.forward_backward(x, y).clone(); g1 = eng.grads.clone(); st1 = eng.state.clone()
    eng.set_params(p0, s0)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5)).cuda()
    l2 = eng.forward_backward(x[perm].contiguous(), y[perm].contiguous()).clone(); g2 = eng.grads.clone()
    torch.cuda.synchronize()
    assert abs(l1.item() - l2.item()) <= 2e-6 * abs(l1.item())
    torch.testing.assert_close(eng.state, st1, rtol=1e-5, atol=1e-7)
    nl = len(eng.layers)
    for li, d in enumerate(eng.layers):
        n = d['cout'] * d['ksize'] ** 2 * d['cin']
        a, b = g1[d['w_off']:d['w_off'] + n], g2[d['w_off']:d['w_off'] + n]
        tol = 1e-4 if li == nl - 1 else 6e-2
        assert (a - b).abs().max().item() <= tol * a.abs().max().item() + 1e-12, (d['darknet_index'], li)
    eng.set_params(p0, s0)


def test_608_gradient_matches_directional_derivative(eng, batch):
    """Forward and backward kernels agree at this size: along d = g / |g|, (L(p + e d) - L(p - e d)) / 2e == |g|."""
    x, y = batch
    p0, s0 = eng.params.clone(), eng.state.clone()
    eng.grads = eng.m = eng.v = None
    eng.forward_backward(x, y)
    g = eng.grads.clone().double()
    gn = g.norm().item()
    d = (g / gn).float()
    ratios = []
    for e in (2e-3, 5e-3):
        eng.set_params(p0 + e * d, s0); lp = eng.forward_backward(x, y).item()
        eng.set_params(p0 - e * d, s0); lm = eng.forward_backward(x, y).item()
        ratios.append((lp - lm) / (2 * e) / gn)
    eng.set_params(p0, s0)
    eng.grads = eng.m = eng.v = None
    assert min(abs(r - 1.0) for r in ratios) < 0.05, ratios


def test_608_default_eval_batch_is_the_largest_the_forward_accepts(eng, batch):
    """evaluate()/test() read ahead default_eval_batch(608) = 40 images: the largest multiple of 8 whose first-layer output (batch x 608^2 x
    32 floats) one 2 GiB buffer descriptor addresses.  That batch runs and agrees per image with the batch-16 forward; 48 images -- the
    default at 416 -- are refused by the C entry point with an error that says why (the context stays usable), and Engine.predict_device
    runs them in parts (inference is per image)."""
    from face_vijnana_yolov3_amd._lib import FvError, lib, ptr
    from face_vijnana_yolov3_amd.face_detection import default_eval_batch
    x, _ = batch
    nb = default_eval_batch(S)
    assert nb == 40
    y16 = eng.predict_device(x).clone()
    xb = torch.cat([x, x, x[:nb - 2 * B]])
    yb = eng.predict_device(xb).clone()
    torch.cuda.synchronize()
    assert tuple(yb.shape) == (nb, 19, 19, 6)
    scale = y16.abs().max().item()
    assert (yb[:B] - y16).abs().max().item() <= 2e-5 * scale and (yb[B:2 * B] - y16).abs().max().item() <= 2e-5 * scale
    del yb
    x48 = torch.cat([x, x, x])
    assert eng.max_infer_batch(S) == 45
    ws = eng._workspace(48, S, False)
    y48 = torch.empty((48, 19, 19, 6), dtype=torch.float32, device='cuda')
    rc = lib().fv_forward_infer(eng.ctx.handle, ptr(eng.params), ptr(eng.state), ptr(x48), 48, S, ptr(ws), ws.numel(), ptr(y48))
    with pytest.raises(FvError, match='2 GiB'):
        eng.ctx.check(rc, 'fv_forward_infer')
    y48 = eng.predict_device(x48)                # 40 + 8
    torch.cuda.synchronize()
    assert tuple(y48.shape) == (48, 19, 19, 6)
    for k in range(3):
        assert (y48[k * B:(k + 1) * B] - y16).abs().max().item() <= 2e-5 * scale
    assert torch.equal(eng.predict_device(x), y16)
