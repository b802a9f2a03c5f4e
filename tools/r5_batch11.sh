#!/bin/bash
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out
rm -f $out/r5_hist.txt
TAG=streams MODE=streams NSTREAMS=6 timeout -k 10 300 python tools/late_context_probe.py 2>&1 | grep "three-scale" | tee -a $out/r5_hist.txt
TAG=streams_hwq8 GPU_MAX_HW_QUEUES=8 MODE=streams NSTREAMS=6 timeout -k 10 300 python tools/late_context_probe.py 2>&1 | grep "three-scale" | tee -a $out/r5_hist.txt
TAG=streams_nooverlap FV_OPTIONS=overlap=0 MODE=streams NSTREAMS=4 timeout -k 10 300 python tools/late_context_probe.py 2>&1 | grep "three-scale" | tee -a $out/r5_hist.txt
