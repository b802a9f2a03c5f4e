#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 300 python tools/yolo3_profile.py > gpurun_out/r2at_yolo3.txt 2>&1; echo "rc=$?"
cat gpurun_out/r2at_yolo3.txt | tail -30
