#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_fused_slots_gpu.py tests/test_fullsize_gpu.py tests/test_net_gpu.py -m gpu -q --tb=short -x > gpurun_out/r2af_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r2af_tests.log
for rep in 1 2; do
for v in base vrow; do
  if [ $v = base ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=tools/_variants/libfv_$v.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-detect --no-loader --steps 10 > gpurun_out/r2af_bench_${v}_$rep.json 2> gpurun_out/r2af_bench_${v}_$rep.err; echo "bench $v $rep rc=$?"
done
done
