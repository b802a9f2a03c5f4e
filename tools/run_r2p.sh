#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_ops_gpu.py tests/test_fused_slots_gpu.py tests/test_net_gpu.py -m gpu -q --tb=short > gpurun_out/r2p_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2p_tests.log
for v in 1 0 1 0; do
  FV_CONV_WAVES8=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-loader --no-detect --steps 20 --profile-steps 1 > gpurun_out/r2p_bench_${v}_$RANDOM.json 2>/dev/null; echo "bench $v rc=$?"
done
