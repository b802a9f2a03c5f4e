#!/bin/bash
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 900 python -m pytest tests/test_fused_slots_gpu.py tests/test_net_gpu.py tests/test_yolov3_gpu.py tests/test_fullsize_gpu.py tests/test_face_detector_gpu.py -x -q -k "small_m or option or forward_infer or forward_base or three_scale_forward or per_image or detect_matches or csv_rows or 608" > $out/r5_b16_tests.log 2>&1 || { tail -40 $out/r5_b16_tests.log; exit 1; }
tail -3 $out/r5_b16_tests.log
rm -f $out/r5_b16_ab.txt
for o in conv_small=1 conv_small=0 conv_small=1 conv_small=0; do FV_OPTIONS=$o TAG=$o timeout -k 10 100 python tools/bs1_shapes.py 2>&1 | grep "^#" | tee -a $out/r5_b16_ab.txt; done
TAG=small FV_OPTIONS=conv_small=1 timeout -k 10 100 python tools/bs1_shapes.py > $out/r5_b16_shapes.txt 2>&1; grep -v amdgpu $out/r5_b16_shapes.txt | head -24 | cut -c1-130
for bs in "2 416" "4 416" "1 608" "1 320"; do for o in conv_small=1 conv_small=0; do FV_OPTIONS=$o TAG=$o timeout -k 10 100 python tools/bs1_shapes.py $bs 2>&1 | grep "^#" | tee -a $out/r5_b16_ab.txt; done; done
