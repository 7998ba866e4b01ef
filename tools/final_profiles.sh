#!/bin/bash
# Everything under profiles/ that a round's numbers come from, taken from ONE tree on the GPU box (run from the repo root):
#   bash tools/final_profiles.sh r04 <commit>
# -> gpurun_out/<tag>_*: the bench line, rocprofv3 kernel stats + a steady-state trace excerpt of the same command, the counter
#    passes (tools/pmc_collect.sh), the BASELINE configs through the public surface with their kernel stats, the beam counters.
# Every text artefact starts with the commit it was taken from.
set -u
TAG=${1:-r06}
COMMIT=${2:-unknown}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
hdr() { echo "# commit $COMMIT, $(date -u +%Y-%m-%dT%H:%MZ), MI355X (gfx950), tools/final_profiles.sh"; }

# 1. the bench line: the defaults (96 timed steps), and the driver's own command
python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
tail -c 600 $O/${TAG}_bench.json | head -c 300; echo
python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_command.json 2> $O/${TAG}_bench_driver_command.err

# 2. kernel stats + trace of the same timed command (without the CPU baseline and the side paths: their kernels are not the timed ones)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- python3 bench.py --no-cpu-baseline --no-side-paths > $O/${TAG}_trace_bench.json 2> $O/${TAG}_trace.log
S=$(ls $O/${TAG}_trace/*/*kernel_stats.csv | head -1); F=$(ls $O/${TAG}_trace/*/*kernel_trace.csv | head -1)
{ hdr; echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-side-paths (4 forwards of 64 clips in flight)"; cat $S; } > $O/${TAG}_kernel_stats.csv
{ hdr; python3 tools/trace_gaps.py $F $O/${TAG}_trace_excerpt.tmp; } > $O/${TAG}_trace_summary.txt
{ hdr; cat $O/${TAG}_trace_excerpt.tmp; } > $O/${TAG}_trace_excerpt.csv; rm -f $O/${TAG}_trace_excerpt.tmp $F

# 3. counter passes (dispatches serialised: every kernel counted running alone)
bash tools/pmc_collect.sh ${TAG} > $O/${TAG}_pmc_collect.log 2>&1
{ hdr; cat $O/${TAG}_pmc_summary.md; } > $O/${TAG}_pmc_summary.tmp && mv $O/${TAG}_pmc_summary.tmp $O/${TAG}_pmc_summary.md
python3 - <<PY
import json
p = "$O/pmc_traffic.json"
d = json.load(open(p))
d["_source"] = "profiles/pmc_traffic.json: builder's counter passes of commit $COMMIT (tools/pmc_collect.sh, dispatches serialised, 64-clip forwards), NOT measured in this run"
json.dump(d, open(p, "w"), indent=1, sort_keys=True)
PY

# 4. the other BASELINE configs through the public surface, and the kernel stats of each
{ hdr; python3 tools/run_configs.py 2 3 4 5 6 2>&1 | grep -v "amdgpu.ids\|Using device\|updated"; } > $O/${TAG}_run_configs.txt
for C in 3 4 5; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_cfg${C} -- python3 tools/run_configs.py $C > $O/${TAG}_cfg${C}.log 2>&1
    S=$(ls $O/${TAG}_cfg${C}/*/*kernel_stats.csv | head -1)
    { hdr; echo "# rocprofv3 --kernel-trace --stats -- python3 tools/run_configs.py $C"; cat $S; } > $O/${TAG}_kernel_stats_config${C}.csv
    if [ $C = 4 ]; then      # what runs beside config 4's whole-device recurrent launches (the turn lock)
        F=$(ls $O/${TAG}_cfg${C}/*/*kernel_trace.csv | head -1)
        { hdr; echo "# tools/exp/overlap_report.py over the kernel trace of tools/run_configs.py 4"; python3 tools/exp/overlap_report.py $F 60; } > $O/${TAG}_config4_overlap.txt
    fi
    rm -f $O/${TAG}_cfg${C}/*/*kernel_trace.csv
done

# 5. beam search counters (one pass per group)
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo $G | cut -d' ' -f1)
    rocprofv3 --pmc $G --output-format csv -d $O/${TAG}_beampmc_${N} -- python3 tools/exp/beam_time.py > $O/${TAG}_beampmc_${N}.log 2>&1
done
{ hdr; echo "# counter passes over tools/exp/beam_time.py, beam_kernel dispatches only"; python3 tools/pmc_summary.py $O/${TAG}_beampmc_* | grep beam_kernel; } > $O/${TAG}_pmc_beam.md
ls -la $O | grep ${TAG}_ | head -40
