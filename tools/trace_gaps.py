#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of the batch pipeline -> per queue: kernels in order with the gap behind the previous kernel of
the same queue; prints the steady-state excerpt (the middle of the run), the mean gap in front of each kernel kind, the
time each kind spends running, and how many compute units' worth of persistent recurrent kernels overlap on average.
    python tools/trace_gaps.py <kernel_trace.csv> [out_excerpt.csv]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
t0 = rows[0][0]
def short(k):
    for key in ("rnn_persist_ring", "rnn_persist_duo", "rnn_persist16", "conv1_f16x3", "conv_f16x3", "gemm_f16x3_kernel<true>", "gemm_f16x3_kernel<false>",
                "gemm_f16x3_wide_kernel<true", "gemm_f16x3_wide_kernel<false", "split_a_kernel", "stft_logmag", "stft_mfma", "normalize", "head_kernel", "greedy_kernel", "beam_kernel", "copyBuffer", "fillBuffer"):
        if key in k:
            return key
    return k[:40]
by_q = collections.defaultdict(list)
for s, e, q, k in rows:
    by_q[q].append((s, e, k))
rr = [r for r in rows if "rnn_persist" in r[3]]
lo, hi = rr[len(rr) // 4][0], rr[3 * len(rr) // 4][0]        # the middle half of the recurrent launches: steady state
gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
for q, ks in by_q.items():
    for i in range(1, len(ks)):
        if lo <= ks[i][0] <= hi:
            gaps[short(ks[i][2])].append((ks[i][0] - ks[i - 1][1]) / 1e3)
            durs[short(ks[i][2])].append((ks[i][1] - ks[i][0]) / 1e3)
print("steady-state window %.1f .. %.1f ms; per kernel kind: launches, mean duration us, mean gap behind the previous kernel of its queue us" % ((lo - t0) / 1e6, (hi - t0) / 1e6))
for k in sorted(durs, key=lambda k: -sum(durs[k])):
    print("  %-50s n %4d  dur %8.1f  gap %8.1f  (total %8.1f ms running, %7.1f ms of gaps)" % (k, len(durs[k]), sum(durs[k]) / len(durs[k]), sum(gaps[k]) / len(gaps[k]), sum(durs[k]) / 1e3, sum(gaps[k]) / 1e3))
# overlap of the persistent recurrent kernels
ring = [(s, e) for s, e, q, k in rows if "rnn_persist" in k and lo <= s <= hi]
ev = sorted([(s, 1) for s, e in ring] + [(e, -1) for s, e in ring])
cur, last, acc = 0, lo, collections.Counter()
for t, dlt in ev:
    acc[cur] += t - last
    cur, last = cur + dlt, t
tot = sum(acc.values())
print("persistent recurrent kernels running at the same time: " + ", ".join("%d: %.0f%%" % (k, 100.0 * v / tot) for k, v in sorted(acc.items())))
# per queue: how long it stands idle between two forwards (from the end of a forward's last kernel to the next forward's first)
for q, ks in sorted(by_q.items()):
    starts = [i for i, (s_, e_, k) in enumerate(ks) if ("stft_logmag" in k or "stft_mfma" in k) and lo <= s_ <= hi]
    if len(starts) < 2:
        continue
    idle, span = [], []
    for a, b in zip(starts[:-1], starts[1:]):
        # the last compute kernel of the forward that started at a: the last kernel before b that is not a copy
        last = max(e_ for s_, e_, k in ks[a:b] if "copyBuffer" not in short(k) and "fillBuffer" not in short(k))
        idle.append((ks[b][0] - last) / 1e6)
        span.append((last - ks[a][0]) / 1e6)
    print("queue %s: %d forwards, %.1f ms each on the device, then %.1f ms idle before the next one starts" % (q, len(idle), sum(span) / len(span), sum(idle) / len(idle)))
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        f.write("# steady-state excerpt of rocprofv3 --kernel-trace over bench.py (tools/trace_gaps.py): start/end in us from the first row\n")
        f.write("start_us,end_us,dur_us,queue,kernel\n")
        ex = [(s, e, q, k) for s, e, q, k in rows if lo <= s <= lo + (hi - lo) * 0.12]
        for s, e, q, k in ex:
            f.write("%.1f,%.1f,%.1f,%s,%s\n" % ((s - ex[0][0]) / 1e3, (e - ex[0][0]) / 1e3, (e - s) / 1e3, q, short(k)))
