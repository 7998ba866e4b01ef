#!/bin/bash
# Instruction counters of the recurrent kernels, one layer alone on the chip (run on the GPU box, from the repo root):
#   bash tools/exp/rnn_inst_counts.sh [H] [B]   -> gpurun_out/rnn_inst_H_B.md (per dispatch means of the persistent kernels)
set -u
H=${1:-800}; B=${2:-64}
export TMPDIR=/tmp
O=gpurun_out/rnn_inst_${H}_${B}
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
    N=$(echo $G | cut -d' ' -f1)
    rocprofv3 --pmc $G --output-format csv -d ${O}_${N} -- python3 tools/exp/ring_layer_time.py $H $B > ${O}_${N}.log 2>&1 || echo "pass $N failed"
done
python3 tools/pmc_summary.py ${O}_* | grep "rnn_" > gpurun_out/rnn_inst_${H}_${B}.md
cat gpurun_out/rnn_inst_${H}_${B}.md
