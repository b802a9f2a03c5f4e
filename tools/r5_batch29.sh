#!/bin/bash
# why does comm mode 'pg' take 44 s per step in the 2-rank gloo rehearsal of bench.py?  GPU_MAX_HW_QUEUES 8 (this round's default) against 4
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out
for q in 4 8; do
  for m in pg wg; do
    s=$(date +%s)
    GPU_MAX_HW_QUEUES=$q FV_COMM_STREAM=$m FV_DIST_BACKEND=gloo FV_BENCH_DEVICE=0 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2951$q bench.py --gpus 2 --steps 2 --warmup 1 --batch 8 --profile-steps 0 > $out/r5_b29_${m}_q$q.json 2> $out/r5_b29_${m}_q$q.err
    rc=$?
    e=$(date +%s)
    echo "queues $q mode $m: rc=$rc wall $((e-s)) s $(python -c "import json; d=json.load(open('$out/r5_b29_${m}_q$q.json')); print('ms_per_step', d['ms_per_step'])" 2>/dev/null)" | tee -a $out/r5_b29.txt
  done
done
