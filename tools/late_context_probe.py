"""Round 5: the three-scale training step (batch 16) in a fresh process, then again after n extra torch streams have been created and used
(MODE=streams NSTREAMS=n) or after a test() loop (MODE=testloop): extra streams alone do not slow a later context down (profiles/r05_late_context_slowdown*.txt)."""
import os, sys, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch, bench
def ts(tag):
    r = bench.three_scale_bench(0, 416, B=16, steps=5)
    print(os.environ.get('TAG'), tag, 'three-scale B16 ms/step', r['ms_per_step'], flush=True)
ts('fresh')
mode = os.environ.get('MODE', 'streams')
if mode == 'testloop':
    bench.test_loop_bench(0, 416, n_img=64)
    ts('after test_loop')
elif mode == 'streams':
    # N extra torch streams, each used once
    x = torch.ones(1 << 20, device='cuda')
    for n in range(int(os.environ.get('NSTREAMS', 1))):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            y = x * 2
        torch.cuda.synchronize()
        ts('after %d extra stream(s)' % (n + 1))
