#!/bin/bash
# round 5, GPU call 5: A/B of three small levers on the 1x1 family: weight-gradient split count, 128x64 tiles for N = 128, early z/addend loads
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
Q="--steps 20 --warmup 10 --no-cpu-baseline --no-detect --no-loader --no-three-scale --no-rccl-rehearsal --profile-steps 0"
rm -f $out/r5_b4_ab.txt
run() {
  tag=$1
  timeout -k 10 200 python bench.py $Q 2> $out/r5_b4_bench.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$tag', d['value'], d['median_ms_per_step'])" | tee -a $out/r5_b4_ab.txt
  timeout -k 10 200 python tools/base_profile.py > $out/r5_b4_shapes_$tag.txt 2>&1
  grep "t1 \|conv1x1\|^sum" $out/r5_b4_shapes_$tag.txt | cut -c1-150
}
run default
FV_WGRAD_MINCH1=12 run minch12
FV_WGRAD_MINCH1=8 run minch8
FV_C1X1_N64=1 run n64
FV_LIB_PATH=$GRAFT_REPO_ROOT/tools/_variants/libfv_early.so run early
run default2
