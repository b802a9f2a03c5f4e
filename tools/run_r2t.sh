#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --no-cpu-baseline --no-detect --steps 10 > gpurun_out/r2t_bench.json 2> gpurun_out/r2t_bench.err; echo "bench rc=$?"
timeout -k 10 300 python -m pytest tests/test_face_detector_gpu.py -m gpu -q --tb=short > gpurun_out/r2t_tests.log 2>&1; echo "tests rc=$?"
bash tools/profile_round.sh r02b
