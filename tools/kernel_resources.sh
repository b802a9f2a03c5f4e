#!/bin/bash
# Register / LDS / occupancy report of every kernel in one csrc file (compiler remarks; no GPU needed):
#   tools/kernel_resources.sh conv_mfma.hip
cd "$(dirname "$0")/../face_vijnana_yolov3_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -c "$1" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|Spill|Occupancy|LDS Size|SGPRs:" | sed 's/^.*remark: //'
