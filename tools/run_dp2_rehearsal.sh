#!/bin/bash
# rehearsal of the N>1 bench path: 2 ranks on the one GPU over gloo (RCCL refuses two ranks on one device)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
FV_DIST_BACKEND=gloo FV_BENCH_DEVICE=0 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 --batch 8 --profile-steps 1 > gpurun_out/dp2_dp2.json 2> gpurun_out/dp2_dp2.err; echo "dp2 rc=$?"
