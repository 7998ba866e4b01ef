#!/usr/bin/env python3
"""Which power reading follows the load?  Prints every hwmon power file of the GPU and rocm-smi's figure, idle and while fp16
GEMMs run (about 4 s each).  power_sources.py"""
import glob, subprocess, threading, time, re
import torch

def readings():
    out = {}
    for f in sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*/power1_*")):
        try:
            out[f.split("/hwmon/")[1]] = open(f).read().strip()
        except OSError as e:
            out[f.split("/hwmon/")[1]] = "unreadable (%s)" % e.__class__.__name__
    try:
        t = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        m = re.findall(r"(Power[^\n]*)", t)
        out["rocm-smi"] = "; ".join(x.strip() for x in m)
    except Exception as e:
        out["rocm-smi"] = repr(e)
    return out

print("idle:", readings())
a = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
b = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
stop = threading.Event()
def burn():
    while not stop.is_set():
        for _ in range(20):
            c = a @ b
        torch.cuda.synchronize()
th = threading.Thread(target=burn); th.start()
for k in range(5):
    time.sleep(1.0)
    print("load +%d s:" % (k + 1), readings(), flush=True)
stop.set(); th.join()
time.sleep(1.0)
print("after:", readings())
