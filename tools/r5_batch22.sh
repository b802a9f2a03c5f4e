#!/bin/bash
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $out/tl_trace
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/tl_trace -- python3 $GRAFT_REPO_ROOT/tools/test_loop_trace.py > $out/r5_tl_trace.log 2>&1 || { tail -20 $out/r5_tl_trace.log; exit 1; }
grep pass $out/r5_tl_trace.log
python3 $GRAFT_REPO_ROOT/tools/test_loop_trace_summary.py $out/tl_trace | tee $out/r5_tl_summary.txt
find $out/tl_trace -name "*.csv" -size +20M -delete
