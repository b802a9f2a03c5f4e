#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_net_gpu.py tests/test_face_detector_gpu.py tests/test_dp_rehearsal_gpu.py tests/test_ops_gpu.py -m gpu -q --tb=short -x > gpurun_out/r2an_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r2an_tests.log
