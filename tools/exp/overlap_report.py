#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of a stream of batches -> does anything run BESIDE the persistent recurrent kernels?
Over the middle half of the recurrent launches: share of wall time with (recurrent only | recurrent + dense | dense only | beam only |
nothing), per kernel kind its launches / mean duration / the share of its running time that lies beside a recurrent kernel, and a
short excerpt of the timeline (start, end, queue, kind).
    python tools/exp/overlap_report.py <kernel_trace.csv> [excerpt rows]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
KEYS = ("rnn_persist_ring4", "rnn_persist_ring", "rnn_persist_duo", "rnn_persist16_pipe", "rnn_persist16", "conv1_f16x3", "conv_f16x3", "gemm_f16x3_wide_kernel<true",
        "gemm_f16x3_wide_kernel<false", "gemm_f16x3_kernel", "split_a_kernel", "stft_mfma", "stft_logmag", "clip_stats", "normalize", "head_kernel", "greedy_kernel",
        "beam_kernel", "copyBuffer", "fillBuffer")
def short(k):
    for key in KEYS:
        if key in k:
            return key
    return k[:40]
rec = [r for r in rows if "rnn_persist" in r[3]]
lo, hi = rec[len(rec) // 4][0], rec[3 * len(rec) // 4][0]
def cls(k):
    if "rnn_persist" in k: return "rec"
    if "beam_kernel" in k: return "beam"
    if "copyBuffer" in k or "fillBuffer" in k: return None
    return "dense"
ev = []
for s, e, q, k in rows:
    c = cls(k)
    if c is None or e < lo or s > hi: continue
    ev.append((max(s, lo), 1, c)); ev.append((min(e, hi), -1, c))
ev.sort()
cur = collections.Counter(); last = lo; acc = collections.Counter()
for t, dlt, c in ev:
    key = ("rec" if cur["rec"] else "") + ("+dense" if cur["dense"] else "") + ("+beam" if cur["beam"] else "") or "nothing"
    acc[key] += t - last
    cur[c] += dlt; last = t
tot = float(sum(acc.values()))
print("window %.1f ms (middle half of the recurrent launches); wall-time shares:" % ((hi - lo) / 1e6))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("   %-22s %5.1f %%" % (k, 100 * v / tot))
recs = [(s, e) for s, e, q, k in rows if "rnn_persist" in k]
def beside(s, e):
    o = 0
    for rs, re_ in recs:
        if re_ <= s: continue
        if rs >= e: break
        o += min(e, re_) - max(s, rs)
    return o
st = collections.defaultdict(lambda: [0, 0, 0])
for s, e, q, k in rows:
    if s < lo or s > hi: continue
    a = st[short(k)]
    a[0] += 1; a[1] += e - s
    if "rnn_persist" not in k: a[2] += beside(s, e)
print("kind: launches, mean ms, total ms, share of its time beside a recurrent kernel")
for k, (n, d, b) in sorted(st.items(), key=lambda kv: -kv[1][1]):
    print("   %-34s %5d  %8.3f  %9.1f   %5.1f %%" % (k, n, d / n / 1e6, d / 1e6, 100.0 * b / max(d, 1)))
nrow = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if nrow:
    print("excerpt (ms from window start): start end dur queue kind")
    ex = [r for r in rows if r[0] >= lo and cls(r[3]) is not None][:nrow]
    for s, e, q, k in ex:
        print("   %9.3f %9.3f %8.3f  q%-3s %s" % ((s - lo) / 1e6, (e - lo) / 1e6, (e - s) / 1e6, q, short(k)))
