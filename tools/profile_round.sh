#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline numbers (run on the GPU box through gpurun):
#   tools/profile_round.sh r01
# writes gpurun_out/<tag>_*: kernel-trace statistics of the serial and the overlapped schedule with
# the bench lines of those very runs, and PMC FETCH_SIZE / WRITE_SIZE in two separate passes
# (counters are never combined with other trace domains).  tools/summarise_profiles.py turns them
# into the files kept under profiles/.
set -e -o pipefail
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$root'); from face_vijnana_yolov3_amd.build import source_fingerprint; print(source_fingerprint())" > "$out/${tag}_fingerprint.txt"
common="--steps 5 --warmup 2 --no-cpu-baseline --no-detect --no-loader --no-rccl-rehearsal"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_serial" -o run -- python3 "$root/bench.py" $common --no-overlap > "$out/${tag}_serial_bench.json" 2> "$out/${tag}_serial.err"
echo "serial done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_overlap" -o run -- python3 "$root/bench.py" $common > "$out/${tag}_overlap_bench.json" 2> "$out/${tag}_overlap.err"
echo "overlap done"
pmc="--steps 1 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-overlap --no-detect --no-loader --no-rccl-rehearsal"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_pmc_fetch" -o run -- python3 "$root/bench.py" $pmc > "$out/${tag}_pmc_fetch_bench.json" 2> "$out/${tag}_pmc_fetch.err"
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/${tag}_pmc_write" -o run -- python3 "$root/bench.py" $pmc > "$out/${tag}_pmc_write_bench.json" 2> "$out/${tag}_pmc_write.err"
echo "write done"
# matrix-core utilisation of every kernel: busy cycles of the MFMA pipe and the active-cycle base (their own pass)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/${tag}_pmc_mfma" -o run -- python3 "$root/bench.py" $pmc > "$out/${tag}_pmc_mfma_bench.json" 2> "$out/${tag}_pmc_mfma.err"
echo "mfma done"
