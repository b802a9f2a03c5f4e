#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_fused_slots_gpu.py -m gpu -q --tb=short -x > gpurun_out/r2al_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r2al_tests.log
timeout -k 10 400 python bench.py --no-cpu-baseline --no-loader --steps 10 > gpurun_out/r2al_bench.json 2> gpurun_out/r2al_bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r2al_bench.err
timeout -k 10 400 python bench.py --no-cpu-baseline --no-loader --no-detect --steps 10 --image-size 608 --batch 16 > gpurun_out/r2al_bench608.json 2> gpurun_out/r2al_bench608.err; echo "bench608 rc=$?"
