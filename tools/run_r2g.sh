#!/bin/bash
# wgrad ablation: address-prep interleave on/off, no atomics, no loads
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
S="3_1_128_256,3_1_256_512,3_1_512_1024,1_1_256_128,3_1_64_128,3_2_32_64,3_1_32_64"
for v in default noprep noatomic noload; do
  if [ $v = default ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=$PWD/tools/_variants/libfv_$v.so; fi
  echo "== $v" >> gpurun_out/r2g_wgrad.txt
  timeout -k 10 200 python tools/layer_bench.py --only $S --reps 10 >> gpurun_out/r2g_wgrad.txt 2>&1
done
echo done
