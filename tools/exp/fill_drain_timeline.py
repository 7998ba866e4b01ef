#!/usr/bin/env python3
"""Where a short run's time goes: the forwards of the LAST recognize_batches call in a rocprofv3 kernel trace of
`bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-paths`, one line per forward (stream, first kernel's start, the
start of its first recurrent window, last kernel's end; ms from the first kernel of the call), and the CUs' busy share over the call
(recurrent windows counted as 50 CUs, every other kernel as the rest of the chip while it runs).

    fill_drain_timeline.py <kernel_trace.csv> [forwards in the call = 10]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
nfw = int(sys.argv[2]) if len(sys.argv) > 2 else 10
is_first = lambda k: "stft_" in k
starts = [i for i, r in enumerate(rows) if is_first(r[3])]
first = starts[-nfw]
t0 = rows[first][0]
call = rows[first:]
by_q = collections.defaultdict(list)
for s, e, q, k in call:
    by_q[q].append((s, e, k))
print("forwards of the last call (%d), ms from its first kernel:" % nfw)
out = []
for q, ks in by_q.items():
    cur = None
    for s, e, k in ks:
        if is_first(k):
            if cur: out.append(cur)
            cur = dict(q=q, start=s, ring=None, end=e, nring=0, ring_ms=0.0, dense_ms=0.0)
        if cur is None: continue
        cur["end"] = max(cur["end"], e)
        if "rnn_persist" in k:
            cur["nring"] += 1; cur["ring_ms"] += (e - s) / 1e6
            if cur["ring"] is None: cur["ring"] = s
        elif "copyBuffer" not in k and "fillBuffer" not in k:
            cur["dense_ms"] += (e - s) / 1e6
    if cur: out.append(cur)
out.sort(key=lambda c: c["start"])
for c in out:
    print("  queue %-3s start %7.2f  first recurrent window %7.2f  end %7.2f  (lasts %6.2f; recurrent %5.2f in %d windows, other kernels %5.2f)" %
          (c["q"], (c["start"] - t0) / 1e6, ((c["ring"] or c["start"]) - t0) / 1e6, (c["end"] - t0) / 1e6, (c["end"] - c["start"]) / 1e6, c["ring_ms"], c["nring"], c["dense_ms"]))
end = max(r[1] for r in call)
print("call: %.2f ms of kernels from first start to last end" % ((end - t0) / 1e6))
# chip occupancy in 2-ms bins
binw = 2e6
nb = int((end - t0) / binw) + 1
ring = [0.0] * nb; dense = [0.0] * nb
for s, e, q, k in call:
    if "copyBuffer" in k or "fillBuffer" in k: continue
    tgt = ring if "rnn_persist" in k else dense
    b0, b1 = int((s - t0) / binw), int((e - t0) / binw)
    for b in range(b0, min(b1, nb - 1) + 1):
        lo, hi = max(s, t0 + b * binw), min(e, t0 + (b + 1) * binw)
        if hi > lo: tgt[b] += (hi - lo) / binw
print("per 2-ms bin: recurrent windows running (mean count) | other kernels running (mean count)")
print("  " + " ".join("%.1f|%.1f" % (a, b) for a, b in zip(ring, dense)))
