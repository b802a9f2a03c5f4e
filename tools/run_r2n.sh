#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
FV_CONV_WAVES8=1 timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -k "conv_forward or tail_split or k_split" > gpurun_out/r2n_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2n_tests.log
S="3_1_128_256,3_1_256_512,3_1_512_1024,1_1_256_128,1_1_512_256,1_1_1024_512,3_1_64_128,3_2_128_256"
for v in 0 1 0 1; do
  echo "== waves8=$v" >> gpurun_out/r2n_lb.txt
  FV_CONV_WAVES8=$v timeout -k 10 200 python tools/layer_bench.py --only $S --reps 10 --scratch-mib 64 >> gpurun_out/r2n_lb.txt 2>&1
done
echo done
