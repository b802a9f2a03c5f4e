#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for v in base vrow; do
  if [ $v = base ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=tools/_variants/libfv_$v.so; fi
  timeout -k 10 200 python tools/layer_bench.py --reps 10 --scratch-mib 256 --only 3_1_128_256,3_1_256_512,1_1_256_128,3_1_64_128,3_1_512_1024 > gpurun_out/r2ag_layers_$v.txt 2>&1; echo "$v rc=$?"
done
