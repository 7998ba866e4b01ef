#!/bin/bash
# Counter passes for the bench workload (run on the GPU box, from the repo root):
#   bash tools/pmc_collect.sh r02
# One rocprofv3 invocation per counter group (TCC slots: FETCH_SIZE and WRITE_SIZE do not fit one pass;
# --pmc is never combined with trace domains other than the kernel trace); the counter passes serialise the
# dispatches, so every kernel -- the default two-in-flight variants included -- is counted running alone.  Raw CSVs land under gpurun_out/<tag>_pmc_<group>/; tools/pmc_report.py turns
# them into profiles/<tag>_pmc_summary.md and profiles/pmc_traffic.json.
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
# (8 timed steps behind a 16-step warm-up call: every forward of the process carries 64 clips -- a step count that is not a multiple of 8
# ends in a round of 32-clip forwards, whose launches would be averaged into the per-launch figures)
CMD="python3 bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-side-paths --no-kernel-sampling"
for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA"; do
    N=$(echo $G | cut -d' ' -f1)
    rocprofv3 --pmc $G --output-format csv -d gpurun_out/${TAG}_pmc_${N} -- $CMD > gpurun_out/${TAG}_pmc_${N}.log 2>&1 || echo "pass $N failed (see gpurun_out/${TAG}_pmc_${N}.log)"
done
python3 tools/pmc_report.py ${TAG} gpurun_out/${TAG}_pmc_* > gpurun_out/${TAG}_pmc_summary.md
cat gpurun_out/${TAG}_pmc_summary.md
