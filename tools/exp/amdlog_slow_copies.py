#!/usr/bin/env python3
"""AMD_LOG_LEVEL=4 stderr of a run -> every hipMemcpyAsync of >= 32 MB with the time between its entry and its return on the calling
thread; for those above 2 ms, the runtime's own lines in between (which path the copy took).   amdlog_slow_copies.py <log> [max lines]"""
import re, sys
ent = re.compile(r":\s*(\d+) us:\s*\[pid:\s*(\d+)\s+tid:\s*(0x[0-9a-f]+)\]\s*(?:\x1b\[\d+m)?\s*hipMemcpyAsync \(\s*(\S+), (\S+), (\d+), (\w+)")
ret = re.compile(r":\s*(\d+) us:\s*\[pid:\s*(\d+)\s+tid:\s*(0x[0-9a-f]+)\]\s*hipMemcpyAsync: Returned")
ts = re.compile(r":\s*(\d+) us:\s*\[pid:\s*(\d+)\s+tid:\s*(0x[0-9a-f]+)\]")
maxl = int(sys.argv[2]) if len(sys.argv) > 2 else 60
open_ = {}
n = 0
for line in open(sys.argv[1], errors="replace"):
    m = ent.search(line)
    if m and int(m.group(6)) >= (32 << 20):
        open_[m.group(3)] = [int(m.group(1)), line.strip()[:260], []]
        continue
    m = ret.search(line)
    if m and m.group(3) in open_:
        t0, first, body = open_.pop(m.group(3))
        dt = int(m.group(1)) - t0
        n += 1
        print("copy %3d: %6.2f ms on the calling thread   %s" % (n, dt / 1e3, first[first.find("hipMemcpyAsync"):][:150]))
        if dt > 2000:
            for b in body[:maxl]:
                print("        | " + b)
        continue
    m = ts.search(line)
    if m and m.group(3) in open_:
        open_[m.group(3)][2].append(line.strip()[:220])
