#!/bin/bash
# Where the waves of the dense kernels (conv2, the x-projection GEMMs) spend their cycles, one 64-clip forward at a time
# (run on the GPU box, from the repo root):  bash tools/exp/dense_wait_counts.sh  -> gpurun_out/dense_wait.md
set -u
export TMPDIR=/tmp
O=gpurun_out/dense_wait
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
    N=$(echo $G | cut -d' ' -f1)
    rocprofv3 --pmc $G --output-format csv -d ${O}_${N} -- python3 tools/exp/kernel_times_1inflight.py 64 > ${O}_${N}.log 2>&1 || echo "pass $N failed"
done
python3 tools/pmc_summary.py ${O}_* | grep -i "conv_f16\|gemm_f16" > gpurun_out/dense_wait.md
cat gpurun_out/dense_wait.md
