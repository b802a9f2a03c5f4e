#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for v in base noload; do
  if [ $v = base ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=tools/_variants/libfv_$v.so; fi
  timeout -k 10 200 python tools/layer_bench.py --reps 10 --scratch-mib 256 > gpurun_out/r2w_layers_$v.txt 2>&1; echo "$v rc=$?"
done
unset FV_LIB_PATH
timeout -k 10 300 python bench.py --no-cpu-baseline --no-detect --no-loader --steps 10 > gpurun_out/r2w_bench.json 2> gpurun_out/r2w_bench.err; echo "bench rc=$?"
