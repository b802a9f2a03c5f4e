#!/bin/bash
# the exit-time SIGSEGV of the rocprofv3-wrapped one-launch forward (gpurun_out/r4_bs1_persist.txt).
# Runs tools/bs1_profile.py under `rocprofv3 --kernel-trace --stats` in the variants that separate the suspects:
#   coop        one-launch forward through hipLaunchCooperativeKernel (what the faulting run used)
#   plain       the same kernel as a plain launch (the round-4 default of the opt-in path)
#   perlayer    per-layer launches (the control: exited cleanly in round 4)
# Each run dumps its library map (BS1_MAPS) so the frames of a fault can be attributed to a library.
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {
  tag=$1
  export BS1_MAPS=$out/r5_exit_$tag.maps
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r5_exit_$tag -o run -- python3 $GRAFT_REPO_ROOT/tools/bs1_profile.py > $out/r5_exit_$tag.txt 2>&1 < /dev/null
  echo "$tag: exit code $?" | tee -a $out/r5_exit_summary.txt
  grep -c SIGSEGV $out/r5_exit_$tag.txt | sed "s/^/$tag: SIGSEGV lines /" | tee -a $out/r5_exit_summary.txt
}
rm -f $out/r5_exit_summary.txt
FV_INFER_PERSIST=1 FV_PERSIST_COOP=1 run coop
FV_INFER_PERSIST=1 FV_PERSIST_COOP=0 run plain
FV_INFER_PERSIST=0 run perlayer
FV_INFER_PERSIST=1 FV_PERSIST_COOP=1 BS1_CLOSE=1 run coop_closed
exit 0
