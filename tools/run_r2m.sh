#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/r2m_bs1 -o run -- python3 $root/tools/bs1_trace.py > $out/r2m.txt 2>&1
echo done
