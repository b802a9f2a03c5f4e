"""The opt-in forms of the small-M (batch-1) inference forward -- `self.model.predict(image)` of detect(), fd.py:899 -- against the
default per-layer launches:
  * fv_set_infer_persist(2): layers 9 .. 51 + head as ONE cooperative launch with the per-layer path's K-split plan: bit-identical;
  * fv_set_infer_persist(1): the same launch with its own plan (more K slices): equal to fp32 rounding, and within the oracle bound;
  * fv_set_fuse_finish1x1(1): split-K finish + the following 1x1 layer in one launch: bit-identical;
  * every device-side wait is bounded and reports (fv_infer_persist_status); a grid the runtime refuses falls back to the launches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from face_vijnana_yolov3_amd.engine import Engine
    e = Engine(0)
    e.init_synthetic(seed=7)
    yield e
    e.ctx.set_infer_persist(0); e.ctx.set_fuse_finish1x1(False)


def _x(B, S, seed=1):
    return torch.rand((B, S, S, 3), generator=torch.Generator().manual_seed(seed)).cuda()


@pytest.mark.parametrize('B,S', [(1, 416), (1, 320), (2, 256), (1, 96), (3, 64)])
def test_one_launch_forward_equals_the_per_layer_launches(eng, B, S):
    x = _x(B, S)
    eng.ctx.set_infer_persist(0); eng.ctx.set_fuse_finish1x1(False)
    y0 = eng.predict_device(x).clone()
    for grid, coop in ((0, False), (256, True), (512, False), (64, True), (0, True)):
        eng.ctx.set_infer_persist_cooperative(coop)          # plain launch behind the library's occupancy check | hipLaunchCooperativeKernel
        eng.ctx.set_infer_persist(2, grid)
        y2 = eng.predict_device(x).clone()
        eng.ctx.infer_persist_status()                       # synchronises; raises if a wait was abandoned on the device
        assert torch.equal(y2, y0), ('plan 2 must be bit-identical', grid)
        eng.ctx.set_infer_persist(1, grid)
        y1 = eng.predict_device(x).clone(); y1b = eng.predict_device(x).clone()
        eng.ctx.infer_persist_status()
        assert torch.equal(y1, y1b), ('repeatable', grid)
        assert (y1 - y0).abs().max().item() <= 2e-5 * max(y0.abs().max().item(), 1e-3), grid
    eng.ctx.set_infer_persist(0); eng.ctx.set_infer_persist_cooperative(False)


def test_one_launch_forward_matches_the_oracle(eng):
    from oracle import net_oracle as no
    from tests.test_net_gpu import _setup, _within
    p64, s64, x, _ = _setup(5, 1, 96)
    y64, _ = no.forward(p64, s64, x, training=False)
    y32, _ = no.forward(p64.float(), s64.float(), x.float(), training=False)
    p0, s0 = eng.params.clone(), eng.state.clone()
    eng.set_params(p64.float(), s64.float())
    try:
        for mode in (1, 2):
            eng.ctx.set_infer_persist(mode)
            y = eng.predict_device(x.float())
            eng.ctx.infer_persist_status()
            _within(y.cpu(), y64, y32, 'one-launch forward, plan %d' % mode)
    finally:
        eng.ctx.set_infer_persist(0)
        eng.set_params(p0, s0)


def test_one_launch_is_only_taken_in_the_small_m_regime_and_reports_its_phases(eng):
    x1, x8 = _x(1, 416), _x(8, 416)
    eng.ctx.set_infer_persist(0)
    y1, y8 = eng.predict_device(x1).clone(), eng.predict_device(x8).clone()
    eng.ctx.set_infer_persist(1)
    eng.ctx.infer_persist_trace(True)
    y1p = eng.predict_device(x1).clone()
    t = eng.ctx.infer_persist_trace(False, read=True)
    assert len(t) == 3 * 44 + 1 and all(b >= a for a, b in zip(t, t[1:])) and 200.0 < t[-1] < 20000.0, (len(t), t[-1])
    assert (y1p - y1).abs().max().item() <= 2e-5 * y1.abs().max().item()
    assert torch.equal(eng.predict_device(x8), y8)           # batch 8: 21 632 pixels at 52 x 52 -- the per-layer path, unchanged
    eng.ctx.infer_persist_status()
    eng.ctx.set_infer_persist(0)


def test_invalid_settings_are_refused(eng):
    from face_vijnana_yolov3_amd._lib import FvError
    with pytest.raises(FvError):
        eng.ctx.set_infer_persist(3)
    with pytest.raises(FvError):
        eng.ctx.set_infer_persist(1, 100)                    # not a multiple of 8
    eng.ctx.set_infer_persist(0)


@pytest.mark.parametrize('B,S', [(1, 416), (1, 608), (2, 256), (1, 96)])
def test_fused_finish_and_1x1_layer_is_bit_identical(eng, B, S):
    x = _x(B, S, seed=3)
    eng.ctx.set_infer_persist(0); eng.ctx.set_fuse_finish1x1(False)
    y0 = eng.predict_device(x).clone()
    f0, _ = eng.predict_base_device(x, with_head=True)
    eng.ctx.set_fuse_finish1x1(True)
    y1 = eng.predict_device(x).clone()
    f1, h1 = eng.predict_base_device(x, with_head=True)
    eng.ctx.set_fuse_finish1x1(False)
    assert torch.equal(y1, y0) and torch.equal(f1, f0) and torch.equal(h1, y0)


def test_a_lost_workgroup_ends_in_an_error_report_not_in_a_hang(monkeypatch):
    """Every device-side wait of the one-launch forward is bounded.  Test hook: workgroup 5 leaves before the third layer barrier
    (FV_PERSIST_TEST_STALL) and the waits give up after 20 000 polls (FV_PERSIST_SPIN; the default is ~1 s): the others time
    out at that barrier, raise the error word and leave, the launch ENDS, fv_infer_persist_status reports it, the NEXT forward
    reports it too if nobody asked, and the context keeps working on the per-layer path."""
    from face_vijnana_yolov3_amd._lib import FvError
    from face_vijnana_yolov3_amd.engine import Engine
    monkeypatch.setenv('FV_PERSIST_TEST_STALL', '5')
    monkeypatch.setenv('FV_PERSIST_SPIN', '20000')
    e = Engine(0)                                     # the environment is read when the context is created
    monkeypatch.delenv('FV_PERSIST_TEST_STALL'); monkeypatch.delenv('FV_PERSIST_SPIN')
    e.init_synthetic(seed=7)
    x = _x(1, 416)
    e.ctx.set_infer_persist(0)
    y0 = e.predict_device(x).clone()
    e.ctx.set_infer_persist(1)
    e.predict_device(x)
    with pytest.raises(FvError, match='abandoned a wait'):
        e.ctx.infer_persist_status()
    e.predict_device(x)                               # again: this time nobody asks for the status ...
    torch.cuda.synchronize()
    with pytest.raises(FvError, match='abandoned a wait'):
        e.predict_device(x)                           # ... and the next call reports it
    e.ctx.set_infer_persist(0)
    assert torch.equal(e.predict_device(x), y0)
