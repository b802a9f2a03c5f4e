#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short > gpurun_out/r2ao_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r2ao_tests.log
