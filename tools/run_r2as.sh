#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/r2as_bench.json 2> gpurun_out/r2as_bench.err; echo "bench rc=$?"
