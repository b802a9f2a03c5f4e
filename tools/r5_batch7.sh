#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
cat > /tmp/ab.py <<'P'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch, bench
from face_vijnana_yolov3_amd.engine import Engine
eng = Engine(0); eng.init_synthetic(7)
x = torch.rand((1, 416, 416, 3)).cuda()
def t():
    for _ in range(5): eng.predict_device(x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): eng.predict_device(x)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 40
print('%s forward b1 %.4f %.4f' % (os.environ.get('TAG'), t(), t()), flush=True)
P
run() { TAG="$*" timeout -k 10 100 env $* python /tmp/ab.py 2>&1 | grep "forward b1" | tee -a $out/r5_ks_sweep.txt; }
rm -f $out/r5_ks_sweep.txt
run FV_KS_TARGET=512
for t in 288 320 352 384 416 448; do for m in 4 8 12; do run FV_KS_TARGET=$t FV_KS_MINSTEPS64=$m; done; done
run FV_KS_TARGET=384 FV_KS_MINSTEPS64=8 FV_KS_MINSTEPS=8
run FV_KS_TARGET=512
