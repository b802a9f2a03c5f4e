#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for rep in 1 2; do
for v in base u2quad; do
  if [ $v = base ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=tools/_variants/libfv_$v.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-detect --no-loader --steps 10 > gpurun_out/r2ac_bench_${v}_$rep.json 2> gpurun_out/r2ac_bench_${v}_$rep.err; echo "bench $v $rep rc=$?"
done
done
