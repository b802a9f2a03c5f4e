#!/bin/bash
# round 5, GPU call: 64-row tiles for the batch-1 forward: tests + detect A/B
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_fused_slots_gpu.py tests/test_net_gpu.py tests/test_face_detector_gpu.py tests/test_yolov3_gpu.py -x -q -k "64_row or option or forward_infer or forward_base or csv_rows or fullsize or three_scale_forward or detect_matches or persistent" > $out/r5_bm64_tests.log 2>&1 || { tail -40 $out/r5_bm64_tests.log; exit 1; }
tail -3 $out/r5_bm64_tests.log
cat > /tmp/ab.py <<'P'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch, bench
from face_vijnana_yolov3_amd.engine import Engine
eng = Engine(0); eng.init_synthetic(7)
x = torch.rand((40, 416, 416, 3)).cuda()
for on in (1, 0, 1, 0):
    eng.ctx.set_option('conv_bm64', on)
    d = bench.detect_bench(eng, x)
    print('conv_bm64=%d batch1_device %.4f batch1_end_to_end %.4f batch40_device %.4f' % (on, d['batch1_device'], d['batch1_end_to_end'], d['batch40_device']), flush=True)
x6 = torch.rand((1, 608, 608, 3)).cuda()
for on in (1, 0):
    eng.ctx.set_option('conv_bm64', on)
    for _ in range(3): eng.predict_device(x6)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): eng.predict_device(x6)
    e1.record(); torch.cuda.synchronize()
    print('608 batch 1 conv_bm64=%d forward %.4f ms' % (on, e0.elapsed_time(e1) / 20))
P
timeout -k 10 300 python /tmp/ab.py 2>&1 | grep conv_bm64 | tee $out/r5_bm64_ab.txt
