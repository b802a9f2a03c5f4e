#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_fused_slots_gpu.py tests/test_net_gpu.py tests/test_ops_gpu.py -m gpu -q -s -x 2>&1 | tail -60 > gpurun_out/r2c_tests.log; echo "tests rc=$?" | tee -a gpurun_out/r2c_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-loader --no-detect --steps 20 > gpurun_out/r2c_bench.json 2> gpurun_out/r2c_bench.err; echo "bench rc=$?"
