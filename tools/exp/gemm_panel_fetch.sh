#!/bin/bash
# (the A/B switches live in the experiments build: make -C danspeech_amd/csrc exp)
export DSMI_LIBRARY=$PWD/danspeech_amd/lib/libdsmi_exp.so
# HBM fetch and time of the split-fp16 GEMMs against the width of the W panel (DSMI_DEBUG_GEMM_PN), bench workload, dispatches serialised by the
# counter pass (run on the GPU box, from the repo root):   bash tools/exp/gemm_panel_fetch.sh "3 4 5 6 8"
export TMPDIR=/tmp
for PN in ${1:-3 4 5 6 8}; do
    O=gpurun_out/gemm_pn_$PN
    DSMI_DEBUG_GEMM_PN=$PN rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-side-paths --no-kernel-sampling > $O.log 2>&1
    echo "PN $PN (FETCH_SIZE x 2 KiB = bytes; mean per dispatch):"
    python3 tools/pmc_summary.py $O | grep gemm_f16x3 | awk '{printf "   %s %s fetch %.2f GB\n", $3, $4, substr($NF,6) * 2048 / 1e9}'
    DSMI_DEBUG_GEMM_PN=$PN python3 bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-side-paths 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('   bench %.3f ms per step; in the pipeline: gemm %.0f us, gemm_l0 %.0f us, ring %.0f us' % (d['ms_per_step'], k['gemm']['avg_us'], k['gemm_l0']['avg_us'], k['rnn_layer_persistent']['avg_us']))"
done
