#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_net_gpu.py -m gpu -q -s --tb=short -k "half_batch or train_step_matches or overlap_equals" > gpurun_out/r2j_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2j_tests.log
for thr in 0 65536 20000 200000; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-loader --no-detect --steps 20 --profile-steps 0 --half-batch-rows $thr > gpurun_out/r2j_bench_$thr.json 2> gpurun_out/r2j_bench_$thr.err; echo "bench $thr rc=$?"
done
