#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_yolov3_gpu.py tests/test_postproc_gpu.py -m gpu -q -s --tb=short -k "three_scale_train_step or iou_pairs" 2>&1 | tail -80 > gpurun_out/r2e_tests.log; echo "tests rc=$?" | tee -a gpurun_out/r2e_tests.log
timeout -k 10 200 python tools/bs1_profile.py > gpurun_out/r2e_bs1.txt 2>&1; echo "bs1 rc=$?"
