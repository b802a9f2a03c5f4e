#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
cat > /tmp/ab.py <<'P'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch
from face_vijnana_yolov3_amd.engine import Engine
eng = Engine(0); eng.init_synthetic(7)
def t(x, n=30):
    for _ in range(4): eng.predict_device(x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): eng.predict_device(x)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = []
for (B, S) in ((1, 416), (2, 416), (3, 416), (4, 416), (6, 416), (8, 416), (1, 608), (2, 608), (1, 320), (1, 224)):
    out.append('%dx%d %.4f' % (B, S, t(torch.rand((B, S, S, 3)).cuda()) / B))
print(os.environ.get('TAG'), 'ms/img:', '  '.join(out), flush=True)
P
run() { TAG="$*" timeout -k 10 100 env $* python /tmp/ab.py 2>&1 | grep "ms/img" | tee -a $out/r5_ks_sizes2.txt; }
rm -f $out/r5_ks_sizes2.txt
for t in 512 480 544 576 512; do run FV_KS_TARGET=$t; done
TAG="floor512" timeout -k 10 100 python tools/bs1_shapes.py > $out/r5_bs1_shapes_floor512.txt 2>&1
