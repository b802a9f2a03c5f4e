#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_yolov3_gpu.py tests/test_postproc_gpu.py tests/test_face_detector_gpu.py tests/test_net_gpu.py -m gpu -q -s -k "three_scale or iou_pairs or csv_rows or fused_bn_backward_switch or decode_nms_matches" 2>&1 | tail -60 > gpurun_out/r2d_tests.log; echo "tests rc=$?" | tee -a gpurun_out/r2d_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-loader --no-detect --steps 20 --profile-steps 0 > gpurun_out/r2d_bench.json 2> gpurun_out/r2d_bench.err; echo "bench rc=$?"
