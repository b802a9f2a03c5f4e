#!/bin/bash
# default eval batch 48: detector tests + the detect section of the bench
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 600 python -m pytest tests/test_face_detector_gpu.py tests/test_jpeg_gpu.py tests/test_three_scale_e2e_gpu.py -x -q > $out/r5_b20_tests.log 2>&1 || { tail -40 $out/r5_b20_tests.log; exit 1; }
tail -3 $out/r5_b20_tests.log
timeout -k 10 500 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-loader --no-three-scale --no-rccl-rehearsal --profile-steps 0 > $out/r5_b20_bench.json 2> $out/r5_b20_bench.err || { tail -30 $out/r5_b20_bench.err; exit 1; }
python -c "
import json; d=json.load(open('$out/r5_b20_bench.json'))
print(json.dumps(d['detect'], indent=1)[:3000])"
