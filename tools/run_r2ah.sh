#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_fused_slots_gpu.py tests/test_fullsize_gpu.py -m gpu -q --tb=short -x > gpurun_out/r2ai_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r2ai_tests.log
for v in base vrow; do
  if [ $v = base ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=tools/_variants/libfv_$v.so; fi
  timeout -k 10 200 python tools/layer_bench.py --reps 10 --scratch-mib 256 --only 3_1_128_256,3_1_256_512,1_1_256_128,3_1_64_128,3_1_512_1024,3_2_128_256 > gpurun_out/r2ai_layers_$v.txt 2>&1; echo "$v rc=$?"
done
unset FV_LIB_PATH
timeout -k 10 300 python bench.py --no-cpu-baseline --no-detect --no-loader --steps 10 > gpurun_out/r2ai_bench.json 2> gpurun_out/r2ai_bench.err; echo "bench rc=$?"
