#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
cd $root
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_fused_slots_gpu.py -x -q > $out/r5_b26_tests.log 2>&1 || { tail -40 $out/r5_b26_tests.log; exit 1; }
tail -2 $out/r5_b26_tests.log
Q="--steps 20 --warmup 10 --no-cpu-baseline --no-detect --no-loader --no-three-scale --no-rccl-rehearsal --profile-steps 0"
rm -f $out/r5_b26_ab.txt
for v in on off on off on off; do
  if [ $v = on ]; then unset FV_OPTIONS; else export FV_OPTIONS=conv_chunk_major=0; fi
  timeout -k 10 200 python3 $root/bench.py $Q 2> $out/r5_b26.err | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['median_ms_per_step'])" | tee -a $out/r5_b26_ab.txt || exit 1
done
