#!/bin/bash
# BN stream passes: first loads in front of the slot prologue + two elements in flight (FV_BN_PIPE), grid 512 / 1024 / 2048
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out
rm -f $out/r5_b19.txt
for v in default bnpipe bnpipe2k bnpipe512 default bnpipe; do
  if [ $v = default ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=$GRAFT_REPO_ROOT/tools/_variants/libfv_$v.so; fi
  echo "== $v" >> $out/r5_b19.txt
  timeout -k 10 200 python tools/bn_bench.py >> $out/r5_b19.txt 2>&1 || exit 1
done
grep "==\|network" $out/r5_b19.txt
export FV_LIB_PATH=$GRAFT_REPO_ROOT/tools/_variants/libfv_bnpipe.so
timeout -k 10 600 python -m pytest tests/test_fused_slots_gpu.py tests/test_ops_gpu.py -x -q -k "bn or slots" > $out/r5_b19_tests.log 2>&1; tail -3 $out/r5_b19_tests.log
