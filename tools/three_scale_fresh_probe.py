"""Round 5: bench.three_scale_bench at batch 40 and 16 in a fresh process (FV_OPTIONS selects kernel options): the yardstick for the late-context slowdown."""
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch, bench
for B in (40, 16):
    r = bench.three_scale_bench(0, 416, B=B, steps=5)
    print(os.environ.get('FV_OPTIONS'), 'B', B, r['value'], r['ms_per_step'], r['frac_of_fp32_mfma_peak'], flush=True)
