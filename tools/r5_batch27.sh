#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
cd $root
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py tests/test_fused_slots_gpu.py -x -q > $out/r5_b27_tests.log 2>&1 || { tail -30 $out/r5_b27_tests.log; exit 1; }; tail -1 $out/r5_b27_tests.log
Q="--steps 20 --warmup 10 --no-cpu-baseline --no-detect --no-loader --no-three-scale --no-rccl-rehearsal --profile-steps 0"
rm -f $out/r5_b27_ab.txt
for v in on off prev on off prev; do
  unset FV_OPTIONS FV_LIB_PATH
  if [ $v = off ]; then export FV_OPTIONS=conv_chunk_major=0; fi
  if [ $v = kmtest ]; then export FV_LIB_PATH=$root/tools/_variants/libfv_kmtest.so; fi
  if [ $v = prev ]; then export FV_LIB_PATH=$root/tools/_variants/libfv_prev.so; fi
  timeout -k 10 200 python3 $root/bench.py $Q 2> $out/r5_b27.err | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['median_ms_per_step'])" | tee -a $out/r5_b27_ab.txt || exit 1
done
