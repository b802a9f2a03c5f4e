#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_fused_slots_gpu.py tests/test_fullsize_gpu.py tests/test_net_gpu.py -m gpu -q --tb=short -x > gpurun_out/r2aj_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r2aj_tests.log
timeout -k 10 200 python tools/layer_bench.py --reps 10 --scratch-mib 256 > gpurun_out/r2aj_layers.txt 2>&1; echo "layers rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-detect --no-loader --steps 10 > gpurun_out/r2aj_bench.json 2> gpurun_out/r2aj_bench.err; echo "bench rc=$?"
