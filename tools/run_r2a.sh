#!/bin/bash
# round-2 GPU pass A: full GPU suite, default bench, side-stream priority A/B
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > gpurun_out/r2a_tests.log; echo "tests rc=$?" | tee -a gpurun_out/r2a_tests.log
timeout -k 10 400 python bench.py > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err; echo "bench rc=$?"
for pr in low high; do
  FV_SIDE_PRIORITY=$pr timeout -k 10 200 python bench.py --no-cpu-baseline --no-detect --no-loader --steps 20 --profile-steps 0 > gpurun_out/r2a_bench_$pr.json 2> gpurun_out/r2a_bench_$pr.err; echo "bench $pr rc=$?"
done
timeout -k 10 200 python bench.py --no-cpu-baseline --no-detect --no-loader --steps 20 --profile-steps 0 > gpurun_out/r2a_bench_dflt.json 2>&1; echo "bench dflt rc=$?"
