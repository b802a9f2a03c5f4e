#!/bin/bash
# round 5: the rocprofv3 evidence (profiles/r05_*) + per-shape tables + phase stamps
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
tools/profile_round.sh r05 > $out/r05_profile_round.log 2>&1; echo "profile_round rc=$?"
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/base_profile.py > $out/r05_base_step_shapes.txt 2>&1; echo "shapes rc=$?"
TAG="round 5" timeout -k 10 100 python tools/bs1_shapes.py > $out/r05_detect_batch1_shapes.txt 2>&1; echo "bs1 rc=$?"
FV_LIB_PATH=$GRAFT_REPO_ROOT/tools/_variants/libfv_stamps.so timeout -k 10 200 python tools/conv_phases.py > $out/r05_conv_phases.txt 2>&1; echo "phases rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r05_bs1 -o run -- python3 $GRAFT_REPO_ROOT/tools/bs1_profile.py > $out/r05_bs1.txt 2>&1; echo "bs1 rocprof rc=$?"
