#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -s --tb=short > gpurun_out/r2f_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2f_tests.log
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err; echo "bench rc=$?"
timeout -k 10 200 python tools/bs1_profile.py > gpurun_out/r2f_bs1.txt 2>&1; echo "bs1 rc=$?"
