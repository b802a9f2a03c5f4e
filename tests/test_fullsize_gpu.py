"""BASELINE-size checks (416x416, batch 40 -- BASELINE.json configs[1]) where the CPU oracle cannot
run in seconds: size-independent properties of the hot path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, S = 40, 416


@pytest.fixture(scope='module')
def eng():
    from face_vijnana_yolov3_amd.engine import Engine
    e = Engine(0)
    e.init_synthetic(seed=7)
    return e


@pytest.fixture(scope='module')
def batch():
    from face_vijnana_yolov3_amd import data
    g = torch.Generator().manual_seed(1234)
    x = torch.rand((B, S, S, 3), generator=g).cuda()
    y = torch.from_numpy(data.synth_gt_batch(B, S, seed=1234)).cuda()
    return x, y


def test_inference_is_per_image_and_deterministic(eng, batch):
    """Inference-mode BN is per-sample: predicting the batch in two parts (different tile counts,
    different tail tiles; the smaller parts take the K-split path for the 13x13 layers, i.e. another
    fp32 summation order) must reproduce the full-batch result to fp32 rounding; repeated calls are
    bit-identical."""
    x, _ = batch
    y = eng.predict_device(x).clone()
    y2 = eng.predict_device(x).clone()
    assert torch.equal(y, y2)
    ya = eng.predict_device(x[:23].contiguous()).clone()
    yb = eng.predict_device(x[23:].contiguous()).clone()
    ys = torch.cat([ya, yb])
    assert (ys - y).abs().max().item() <= 2e-5 * y.abs().max().item()
    assert torch.isfinite(y).all() and tuple(y.shape) == (B, 13, 13, 6)


def test_train_step_batch_permutation_invariance(eng, batch):
    """Batch statistics, the loss and every gradient are symmetric in the batch order."""
    x, y = batch
    p0, s0 = eng.params.clone(), eng.state.clone()
    eng.grads = eng.m = eng.v = None
    l1 = eng.forward_backward(x, y).clone(); g1 = eng.grads.clone(); st1 = eng.state.clone()
    eng.set_params(p0, s0)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
    l2 = eng.forward_backward(x[perm].contiguous(), y[perm].contiguous()).clone(); g2 = eng.grads.clone()
    torch.cuda.synchronize()
    assert abs(l1.item() - l2.item()) <= 2e-6 * abs(l1.item())
    torch.testing.assert_close(eng.state, st1, rtol=1e-5, atol=1e-7)
    # different summation order (tiles, atomics, BN partials): per layer, relative to the layer's
    # gradient scale.  The randomly initialised network is ill-conditioned: a 1e-7 relative input
    # perturbation already moves every base-layer gradient by ~1e-2 (LeakyReLU sign flips;
    # tools/sensitivity_check.py), while identical inputs reproduce to 3e-6 (tools/determinism_check.py).
    # So: tight bound on the head (no BN / leaky behind it), the measured conditioning elsewhere.
    nl = len(eng.layers)
    for li, d in enumerate(eng.layers):
        n = d['cout'] * d['ksize'] ** 2 * d['cin']
        a, b = g1[d['w_off']:d['w_off'] + n], g2[d['w_off']:d['w_off'] + n]
        tol = 1e-4 if li == nl - 1 else 6e-2
        assert (a - b).abs().max().item() <= tol * a.abs().max().item() + 1e-12, (d['darknet_index'], li)
    eng.set_params(p0, s0)


def test_gradient_matches_directional_derivative(eng, batch):
    """Forward and backward kernels agree at full size: along the normalised gradient direction d,
    (L(p + e d) - L(p - e d)) / 2e  ==  <g, d> = |g|  (training-mode forward, fp32)."""
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
    assert min(abs(r - 1.0) for r in ratios) < 0.05, ratios


def test_training_reduces_the_loss(eng, batch):
    """A few Adam steps on one batch drive the MSE down (the end-to-end sanity the reference's
    training log would show)."""
    x, y = batch
    p0, s0 = eng.params.clone(), eng.state.clone()
    eng.grads = eng.m = eng.v = None
    eng.iterations = 0
    losses = []
    for _ in range(6):
        losses.append(eng.train_on_batch(x, y, 1e-4, 0.99, 0.99).item())
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    eng.set_params(p0, s0)
    eng.grads = eng.m = eng.v = None
    eng.iterations = 0
