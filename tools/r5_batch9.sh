#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
for t in 512 448 400 352 320; do TAG="FV_KS_TARGET=$t" FV_KS_TARGET=$t timeout -k 10 100 python tools/bs1_shapes.py > $out/r5_bs1_shapes_$t.txt 2>&1; head -1 $out/r5_bs1_shapes_$t.txt; done
