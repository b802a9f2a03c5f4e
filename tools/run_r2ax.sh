#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_fused_slots_gpu.py -m gpu -q --tb=short > gpurun_out/r2ax_tests.log 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/r2ax_tests.log
timeout -k 10 200 python tools/layer_bench.py --reps 10 --scratch-mib 256 --only 3_2_32_64,3_1_32_64 > gpurun_out/r2ax_layers.txt 2>&1; echo "layers rc=$?"
tail -4 gpurun_out/r2ax_layers.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --no-detect --no-loader --steps 10 > gpurun_out/r2ax_bench.json 2> gpurun_out/r2ax_bench.err; echo "bench rc=$?"
