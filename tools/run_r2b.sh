#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -s 2>&1 | tail -150 > gpurun_out/r2b_tests.log; echo "tests rc=$?" | tee -a gpurun_out/r2b_tests.log
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err; echo "bench rc=$?"
