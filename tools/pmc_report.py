#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs (tools/pmc_collect.sh) -> a markdown table per kernel (mean counter value per dispatch) on
stdout, and profiles/pmc_traffic.json = {bench kernel kind: {"bytes_per_launch", "fetch_bytes", "write_bytes",
"l2_hit", "mfma_busy"}} that bench.py folds into its JSON line.

Corrections, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE
counts 64 B per 128-B request of a wide coalesced read, so it is doubled before it is compared with a byte count;
WRITE_SIZE is exact for 16-byte-per-lane streaming stores."""
import collections
import csv
import glob
import json
import os
import sys

KINDS = [("rnn_persist", "rnn_layer_persistent"), ("gemm_f16x3_kernel<true", "gemm_l0"), ("gemm_f16x3_kernel<false", "gemm"),
         ("gemm_f16x3_wide_kernel<true", "gemm_l0"), ("gemm_f16x3_wide_kernel<false", "gemm"),
         ("conv_f16x3_kernel", "conv2"), ("conv1_f16x3", "conv1"), ("conv_kernel<0>", "conv1"), ("head_kernel", "head"),
         ("stft_logmag", "stft"), ("stft_mfma", "stft"), ("split_a_kernel", "split_a"), ("greedy_kernel", "greedy"), ("normalize_kernel", "normalize"),
         ("clip_stats", "clip_stats")]


def kind_of(name):
    for pat, kind in KINDS:
        if pat in name:
            return kind
    return None


def main():
    tag, dirs = sys.argv[1], sys.argv[2:]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            for r in csv.DictReader(open(f)):
                k = kind_of(r["Kernel_Name"])
                if k:
                    vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    print("# %s: rocprofv3 --pmc, mean per dispatch (bench.py defaults: cfgA, 32 x 10 s, two batches in flight -- dispatches serialised by the counter passes)\n" % tag)
    counters = sorted({c for k in vals for c in vals[k]})
    print("| kernel | " + " | ".join(counters) + " |")
    print("|---|" + "---|" * len(counters))
    for k in sorted(vals):
        mean = {c: sum(v) / len(v) for c, v in vals[k].items()}
        print("| %s | " % k + " | ".join(("%.4g" % mean[c]) if c in mean else "" for c in counters) + " |")
        e = {}
        if "FETCH_SIZE" in mean:
            e["fetch_bytes"] = mean["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in mean:
            e["write_bytes"] = mean["WRITE_SIZE"] * 1024
        if "fetch_bytes" in e and "write_bytes" in e:
            e["bytes_per_launch"] = e["fetch_bytes"] + e["write_bytes"]
        if "TCC_HIT_sum" in mean and mean.get("TCC_HIT_sum", 0) + mean.get("TCC_MISS_sum", 0) > 0:
            e["l2_hit"] = round(mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"]), 4)
        # SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs (= cycles per MFMA x SQ_INSTS_MFMA: 32 for 32x32x16 f16,
        # 16 for 16x16x32 f16, 64 for 32x32x2 f32); SQ_BUSY_CYCLES sums the busy cycles of the 32 shader engines (32 SIMDs each);
        # GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles.  MFMA pipe utilisation over the dispatch, both ways:
        if mean.get("SQ_BUSY_CYCLES"):
            e["mfma_busy"] = round(mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (mean["SQ_BUSY_CYCLES"] * 32.0), 4)
        if mean.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy_grbm"] = round(mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (mean["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
        out[k] = e
    print("\nDerived (FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 = bytes per launch; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES), mfma_busy_grbm = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8)):\n")
    print("```json\n" + json.dumps(out, indent=1, sort_keys=True) + "\n```")
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open(os.path.join("gpurun_out", "pmc_traffic.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
