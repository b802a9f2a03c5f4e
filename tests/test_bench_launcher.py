"""`python bench.py --gpus N` must start its own ranks: fresh children through torch.distributed.run
before the parent touches a GPU, rank 0's JSON line relayed on stdout, a failing child -> non-zero exit.  CPU rehearsal with
two gloo ranks (`--rendezvous-only`: rendezvous + one all-reduce; no compute)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['FV_DIST_BACKEND'] = 'gloo'
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + extra, env=env, capture_output=True, text=True, timeout=300)


def test_self_launch_two_ranks_relays_one_json_line():
    r = _run(['--gpus', '2', '--rendezvous-only'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d['metric'] == 'rendezvous-only' and d['n_gpus'] == 2 and d['value'] == 3.0 and d['backend'] == 'gloo'
    assert 'starting 2 ranks' in r.stderr and 'torch.distributed.run' in r.stderr


def test_self_launch_reports_a_failing_rank():
    r = _run(['--gpus', '2', '--rendezvous-only'], {'FV_DIST_BACKEND': 'no-such-backend'})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]


def test_spawn_flag_goes_through_the_launcher_at_one_rank():
    r = _run(['--gpus', '1', '--spawn', '--rendezvous-only'])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip())
    assert d['n_gpus'] == 1 and d['value'] == 1.0


def test_face_detection_main_starts_num_gpus_ranks(tmp_path, monkeypatch):
    """fd_conf.multi_gpu + num_gpus > 1 (fd.py:358-371): main() starts the ranks itself and never constructs a FaceDetector in
    the parent; inside a rank (WORLD_SIZE set) it does not launch again."""
    from face_vijnana_yolov3_amd import face_detection, parallel
    monkeypatch.chdir(tmp_path)
    conf = {'mode': 'train', 'multi_gpu': True, 'num_gpus': 4}
    json.dump({'fd_conf': conf, 'fi_conf': {}}, open('face_vijnana_yolov3.json', 'w'))
    calls = []
    monkeypatch.setattr(parallel, 'launch_ranks', lambda n, target, *a, **k: calls.append((n, list(target))) or 7)
    monkeypatch.setattr(face_detection, 'FaceDetector', lambda conf: (_ for _ in ()).throw(AssertionError('parent built a detector')))
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    try:
        face_detection.main()
        raise AssertionError('main() returned')
    except SystemExit as e:
        assert e.code == 7
    assert calls == [(4, ['-m', 'face_vijnana_yolov3_amd.face_detection'])]
    # a rank: builds the detector (here: the stand-in raises), does not launch
    monkeypatch.setenv('WORLD_SIZE', '4')
    try:
        face_detection.main()
        raise AssertionError('no detector built')
    except AssertionError as e:
        assert 'parent built a detector' in str(e)
    assert len(calls) == 1
