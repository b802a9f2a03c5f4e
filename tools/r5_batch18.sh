#!/bin/bash
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_fused_slots_gpu.py tests/test_net_gpu.py -x -q -k "not overfits and not bucketed and not 2x2 and not 608" > $out/r5_b18_tests.log 2>&1 || { tail -40 $out/r5_b18_tests.log; exit 1; }
tail -3 $out/r5_b18_tests.log
Q="--steps 20 --warmup 10 --no-cpu-baseline --no-detect --no-loader --no-three-scale --no-rccl-rehearsal --profile-steps 0"
rm -f $out/r5_b18_ab.txt
for v in new prev new prev new prev; do
  if [ $v = new ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=$GRAFT_REPO_ROOT/tools/_variants/libfv_prev.so; fi
  timeout -k 10 200 python bench.py $Q 2> $out/r5_b18.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['median_ms_per_step'])" | tee -a $out/r5_b18_ab.txt
done
