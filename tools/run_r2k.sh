#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/r2k_counters.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $out/r2k_sq -o run -- python3 $root/tools/layer_bench.py --only 3_1_128_256,3_1_512_1024,1_1_256_128 --reps 3 > $out/r2k_sq.txt 2>&1
echo "sq done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VALU --output-format csv -d $out/r2k_sq2 -o run -- python3 $root/tools/layer_bench.py --only 3_1_128_256,3_1_512_1024,1_1_256_128 --reps 3 > $out/r2k_sq2.txt 2>&1
echo "sq2 done"
