#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_fused_slots_gpu.py -m gpu -q --tb=short -k "first_layer or conv_slots" > gpurun_out/r2h_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2h_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-loader --no-detect --steps 20 > gpurun_out/r2h_bench.json 2> gpurun_out/r2h_bench.err; echo "bench rc=$?"
