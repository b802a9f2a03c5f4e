#!/bin/bash
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out
rm -f $out/r5_hist3.txt
run() { tag=$1; shift; timeout -k 10 400 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-loader 2> $out/r5_hist3.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); t=d['three_scale_train']; print('$tag', 'B40', t['value'], t['ms_per_step'], 'B16', t['batch16']['value'], t['batch16']['ms_per_step'], 'hist', d['process_history_check']['ratio'], 'cfg5', d['config5_608_bs16']['median_ms_per_step'])" | tee -a $out/r5_hist3.txt; }
run C_base
GPU_MAX_HW_QUEUES=8 run C_hwq8
FV_SIDE_PRIORITY=default run C_sideprio_default
FV_OPTIONS=overlap=0 run C_nooverlap
