#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
hdr() { echo "# commit a3056a9 (+ tools/pmc_collect.sh: 8 timed steps), $(date -u +%Y-%m-%dT%H:%MZ), MI355X (gfx950), tools/pmc_collect.sh"; }
bash tools/pmc_collect.sh r06 > gpurun_out/r06_pmc_collect.log 2>&1
{ hdr; cat gpurun_out/r06_pmc_summary.md; } > gpurun_out/r06_pmc_summary.tmp && mv gpurun_out/r06_pmc_summary.tmp gpurun_out/r06_pmc_summary.md
python3 - <<PY
import json
p = "gpurun_out/pmc_traffic.json"
d = json.load(open(p))
d["_source"] = "profiles/pmc_traffic.json: builder's counter passes of commit a3056a9 (tools/pmc_collect.sh, dispatches serialised, 64-clip forwards), NOT measured in this run"
json.dump(d, open(p, "w"), indent=1, sort_keys=True)
print({k: round(v["bytes_per_launch"] / 1e9, 3) for k, v in d.items() if isinstance(v, dict)})
print("ring fetch", d["rnn_layer_persistent"]["fetch_bytes"] / 1e9)
PY
