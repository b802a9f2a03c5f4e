#!/bin/bash
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 700 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-three-scale --no-rccl-rehearsal --profile-steps 0 > $out/r5_b21_bench.json 2> $out/r5_b21_bench.err || { tail -30 $out/r5_b21_bench.err; exit 1; }
python -c "
import json; d=json.load(open('$out/r5_b21_bench.json'))
t=d['detect']; 
for k in ('test_loop','test_loop_three_scale_head'):
    v=dict(t[k]); v.pop('path',None); print(k, json.dumps(v))
print(json.dumps(d['loader_inclusive'])[:400])"
