#!/bin/bash
# usage: run_bs1.sh tag  (env selects the variant)
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -o run -- python3 $GRAFT_REPO_ROOT/tools/bs1_profile.py > $out/$tag.txt 2>&1 < /dev/null
grep "bs1 predict" $out/$tag.txt
f=$out/$tag/run_kernel_stats.csv
if [ -f "$f" ]; then cut -d, -f1-4 "$f" | cut -c1-110 | head -9; fi
