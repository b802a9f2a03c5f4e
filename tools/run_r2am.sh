#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_yolov3_gpu.py -m gpu -q --tb=short -x > gpurun_out/r2am_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r2am_tests.log
timeout -k 10 400 python bench.py --no-cpu-baseline --no-loader --steps 5 > gpurun_out/r2am_bench.json 2> gpurun_out/r2am_bench.err; echo "bench rc=$?"
