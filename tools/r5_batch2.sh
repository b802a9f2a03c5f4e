#!/bin/bash
# round 5, GPU call 3: persistent 1x1 kernel -- parity tests, per-shape table, headline A/B
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_fused_slots_gpu.py -x -q -k "persistent or option or eight_wave or bnred" > $out/r5_p1_tests.log 2>&1 || { tail -30 $out/r5_p1_tests.log; exit 1; }
tail -3 $out/r5_p1_tests.log
Q="--steps 20 --warmup 10 --no-cpu-baseline --no-detect --no-loader --no-three-scale --no-rccl-rehearsal --profile-steps 0"
for opt in "conv1x1_persist=1" "conv1x1_persist=0" "conv1x1_persist=1" "conv1x1_persist=0"; do
  FV_OPTIONS=$opt timeout -k 10 200 python bench.py $Q 2> $out/r5_p1_bench.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$opt', d['value'], d['median_ms_per_step'])" | tee -a $out/r5_p1_ab.txt
done
timeout -k 10 300 python tools/base_profile.py > $out/r5_p1_shapes.txt 2>&1
grep -i "1x1\|K128 \|K256 \|K512 \|K1024 \|K64 \|sum" $out/r5_p1_shapes.txt | head -40
