#!/bin/bash
# the default bench line, as the driver runs it (writes gpurun_out/default_bench.json)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/default_bench.json 2> gpurun_out/default_bench.err; echo "bench rc=$?"
