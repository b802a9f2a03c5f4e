#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short > gpurun_out/r2aq_tests.log 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/r2aq_tests.log
timeout -k 10 200 python tools/layer_bench.py --reps 10 --scratch-mib 256 --only 3_2_32_64,3_1_32_64 > gpurun_out/r2aq_layers.txt 2>&1; echo "layers rc=$?"
cat gpurun_out/r2aq_layers.txt | tail -4
