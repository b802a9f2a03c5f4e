#!/bin/bash
# rocprofv3 kernel statistics of the batch-1 detect path (fv_forward_infer + fv_decode_nms, BASELINE metric 2):
#   tools/profile_bs1.sh r03   ->  gpurun_out/r03_bs1/ (copy the *kernel_stats.csv into profiles/)
set -e -o pipefail
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_bs1" -o run -- python3 "$root/tools/bs1_profile.py" > "$out/${tag}_bs1.txt" 2> "$out/${tag}_bs1.err"
cat "$out/${tag}_bs1.txt"
