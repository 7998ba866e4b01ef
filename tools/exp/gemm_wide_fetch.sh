#!/bin/bash
# (the A/B switches live in the experiments build: make -C danspeech_amd/csrc exp)
export DSMI_LIBRARY=$PWD/danspeech_amd/lib/libdsmi_exp.so
# HBM fetch of the split-fp16 GEMMs: the 128 x 256 tile (pairs of n-tiles per W panel: DSMI_DEBUG_GEMM_PN = n-tiles, halved) against the 128 x 128
# tile, bench workload, dispatches serialised by the counter pass (on the GPU box, from the repo root):  bash tools/exp/gemm_wide_fetch.sh
export TMPDIR=/tmp
run() {   # name, env...
    local name=$1; shift
    local O=gpurun_out/gemm_fetch_$name
    env "$@" true
    ( export "$@"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-side-paths --no-kernel-sampling > $O.log 2>&1 )
    echo "$name (FETCH_SIZE x 2 KiB = bytes; mean per dispatch):"
    python3 tools/pmc_summary.py $O | grep gemm_f16x3 | awk '{printf "   %s %s fetch %.2f GB\n", $3, $4, substr($NF,6) * 2048 / 1e9}'
}
run tile128_pn5 DSMI_DEBUG_GEMM_WIDE=0
run tile256_pn4 DSMI_DEBUG_GEMM_WIDE=1 DSMI_DEBUG_GEMM_PN=4
run tile256_pn6 DSMI_DEBUG_GEMM_WIDE=1 DSMI_DEBUG_GEMM_PN=6
run tile256_pn8 DSMI_DEBUG_GEMM_WIDE=1 DSMI_DEBUG_GEMM_PN=8
