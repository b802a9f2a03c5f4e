#!/bin/bash
# round 5, GPU call 4: wave priority of prologue / epilogue against the K loop (s_setprio), A/B of whole builds; phase stamps
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
Q="--steps 20 --warmup 10 --no-cpu-baseline --no-detect --no-loader --no-three-scale --no-rccl-rehearsal --profile-steps 0"
rm -f $out/r5_prio_ab.txt
for v in default prio_pe prio_k default prio_pe; do
  if [ $v = default ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=$GRAFT_REPO_ROOT/tools/_variants/libfv_$v.so; fi
  timeout -k 10 200 python bench.py $Q 2> $out/r5_prio_bench.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['median_ms_per_step'])" | tee -a $out/r5_prio_ab.txt
done
for v in stamps stamps_pe; do
  FV_LIB_PATH=$GRAFT_REPO_ROOT/tools/_variants/libfv_$v.so timeout -k 10 200 python tools/conv_phases.py > $out/r5_phases_$v.txt 2>&1
  echo "== $v"; grep -v "resident\|amdgpu.ids" $out/r5_phases_$v.txt | cut -c1-330
done
