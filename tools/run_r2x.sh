#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for v in sameaddr; do
  export FV_LIB_PATH=tools/_variants/libfv_$v.so
  timeout -k 10 200 python tools/layer_bench.py --reps 10 --scratch-mib 256 > gpurun_out/r2x_layers_$v.txt 2>&1; echo "$v rc=$?"
done
