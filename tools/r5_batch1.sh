#!/bin/bash
# round 5, GPU call 2: exit-fault diagnosis, conv phase stamps, side-stream CU mask A/B, one-launch forward after the launch-bounds fix
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
tools/variants_src/exit_fault_diag.sh > $out/r5_exit_diag.log 2>&1
cat $out/r5_exit_summary.txt
echo "== conv phases"
FV_LIB_PATH=tools/_variants/libfv_stamps.so timeout -k 10 200 python tools/conv_phases.py > $out/r5_conv_phases.txt 2>&1
tail -30 $out/r5_conv_phases.txt
echo "== detect A/B"
timeout -k 10 300 python tools/variants_src/detect_ab.py > $out/r5_detect_ab.txt 2>&1
cat $out/r5_detect_ab.txt | grep batch1
echo "== CU mask A/B"
Q="--steps 20 --warmup 10 --no-cpu-baseline --no-detect --no-loader --no-three-scale --no-rccl-rehearsal --profile-steps 0"
for cus in 0 240 224 192 0; do
  FV_SIDE_CUS=$cus timeout -k 10 200 python bench.py $Q 2> $out/r5_cumask_$cus.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('FV_SIDE_CUS=$cus', d['value'], d['median_ms_per_step'])" | tee -a $out/r5_cumask.txt
done
