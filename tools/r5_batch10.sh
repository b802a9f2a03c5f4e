#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_face_detector_gpu.py tests/test_jpeg_gpu.py tests/test_three_scale_e2e_gpu.py -x -q > $out/r5_b10_tests.log 2>&1 || { tail -40 $out/r5_b10_tests.log; exit 1; }
tail -3 $out/r5_b10_tests.log
cat > /tmp/tl.py <<'P'
import os, sys, json
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import bench
print(json.dumps(bench.test_loop_bench(0, 416), indent=1))
print(json.dumps(bench.test_loop_bench(0, 416, head='three_scale'), indent=1))
P
timeout -k 10 400 python /tmp/tl.py 2>&1 | grep -v amdgpu | tee $out/r5_b10_testloop.txt
