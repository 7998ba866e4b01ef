#!/usr/bin/env python3
"""H2D rate of 80 MB pinned uploads over the first seconds of a process, idle and beside compute: does the link start slow?"""
import time, torch
buf = torch.empty(82 << 20, dtype=torch.uint8).pin_memory()
dev = torch.empty(82 << 20, dtype=torch.uint8, device="cuda")
s = torch.cuda.Stream()
a = torch.randn(4096, 4096, device="cuda")
t_start = time.perf_counter()
for phase, busy in (("idle", False), ("beside matmuls", True), ("idle again", False)):
    for i in range(12):
        if busy:
            for _ in range(20):
                a = torch.tanh(a @ a * 1e-3)
        t0 = time.perf_counter()
        with torch.cuda.stream(s):
            dev.copy_(buf, non_blocking=True)
        s.synchronize()
        dt = time.perf_counter() - t0
        print("%5.2f s %-15s upload %6.2f ms = %5.1f GB/s" % (time.perf_counter() - t_start, phase, dt * 1e3, 0.086 / dt), flush=True)
        if not busy:
            time.sleep(0.15)
    torch.cuda.synchronize()
