#!/bin/bash
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 900 python -m pytest tests/test_fused_slots_gpu.py tests/test_net_gpu.py tests/test_yolov3_gpu.py tests/test_fullsize_gpu.py -x -q -k "small_m or option or forward_infer or forward_base or three_scale_forward or per_image" > $out/r5_b15_tests.log 2>&1 || { tail -40 $out/r5_b15_tests.log; exit 1; }
tail -3 $out/r5_b15_tests.log
for o in conv1x1_small=1 conv1x1_small=0 conv1x1_small=1 conv1x1_small=0; do FV_OPTIONS=$o TAG=$o timeout -k 10 100 python tools/bs1_shapes.py 2>&1 | grep "^#" | tee -a $out/r5_b15_ab.txt; done
TAG=small FV_OPTIONS=conv1x1_small=1 timeout -k 10 100 python tools/bs1_shapes.py > $out/r5_b15_shapes.txt 2>&1; grep "small\|32,4,1\|ks8" $out/r5_b15_shapes.txt | cut -c1-130
