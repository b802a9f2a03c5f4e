#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_fused_slots_gpu.py tests/test_ops_gpu.py -m gpu -q --tb=short > gpurun_out/r2r_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2r_tests.log
timeout -k 10 200 python bench.py --no-cpu-baseline --no-loader --no-detect --steps 20 --profile-steps 2 > gpurun_out/r2r_bench.json 2>/dev/null; echo "bench rc=$?"
