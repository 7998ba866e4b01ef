#!/usr/bin/env python3
"""Joules per unit of work for the kernel classes of the bench pipeline (cfgA, 64-clip forwards, 10 s clips), board power sampled
beside each loop (bench.PowerSampler: the device's hwmon file, or rocm-smi):

    idle                      the board with nothing running
    recurrent layers          dsmi_rnn_layer (x-projection GEMM + the persistent recurrent kernel) of layer 1 on four handles / streams,
                              as the pipeline keeps them: W, J per 64-clip layer, and the share of kernel time that is the recurrent kernel
    conv stack                dsmi_conv_stack (conv1 + conv2: dense only) on four handles / streams: W, J per 64-clip stack
    whole pipeline            Recognizer.recognize_batches as bench.py times it: W, J per 32-clip batch

    energy_by_class.py [seconds per loop = 4] [DSMI_RNN_KERNEL for a second pass of the recurrent loop, e.g. ring8]
The figures above idle are what a cut in one class is worth in joules per batch: 5 layers + 1 conv stack + STFT/head/decode per 64 clips."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from danspeech_amd import _native, synthetic as syn

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
ALT = sys.argv[2] if len(sys.argv) > 2 else None
H, B, T = 800, 64, 1001
To = (T + 1) // 2
cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=H, rnn_layers=5, bidirectional=True, context=20)
sd = syn.make_state_dict(2, "gru", H, 5, seed=0, **syn.TALKATIVE)


def sampled(fn, secs):
    """fn(deadline) runs work until the deadline and returns the units it did -> (watts, seconds, units, samples)"""
    torch.cuda.synchronize()
    with bench.PowerSampler(0) as pw:
        t0 = time.perf_counter()
        units = fn(t0 + secs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return pw.mean_watts(), dt, units, pw.count(), pw.source


def report(name, w, dt, units, n, idle, unit):
    if w is None:
        print("%-28s no power reading" % name)
        return
    print("%-28s %7.1f W (%d samples) over %.2f s; %d %s: %.3f J each, %.3f J above idle (%.1f W idle); %.3f ms each"
          % (name, w, n, dt, units, unit, w * dt / max(units, 1), (w - idle) * dt / max(units, 1), idle, dt / max(units, 1) * 1e3), flush=True)


w_idle, dt, _, n, src = sampled(lambda dl: time.sleep(max(dl - time.perf_counter(), 0)) or 0, 2.0)
print("power source: %s; idle %.1f W (%d samples)" % (src, w_idle or -1, n))
w_idle = w_idle or 0.0


def recurrent_loop(kernel):
    if kernel:
        os.environ["DSMI_RNN_KERNEL"] = kernel
    else:
        os.environ.pop("DSMI_RNN_KERNEL", None)
    models = [_native.NativeModel(cfg, sd) for _ in range(4)]
    streams = [torch.cuda.Stream() for _ in range(4)]
    lens = np.full(B, To, dtype=np.int32)
    xs = [torch.randn(To, B, H, device="cuda") * 0.1 for _ in range(4)]
    for m in models:
        m.set_inflight(4)
        m.set_profiling(2)

    def work(deadline):
        units = 0
        while time.perf_counter() < deadline:
            for k in range(4):
                with torch.cuda.stream(streams[k]):
                    models[k].rnn_layer(1, xs[k], lens)
            units += 4
            if units % 16 == 0:
                torch.cuda.synchronize()
        return units
    work(time.perf_counter() + 0.5)
    torch.cuda.synchronize()
    for m in models:
        m.reset_kernel_stats()
    w, dt, units, n, _ = sampled(work, SECS)
    ks = models[0].kernel_stats()
    ring, gemm = ks.get("rnn_layer_persistent"), ks.get("gemm")
    report("recurrent layers (%s)" % (kernel or "default"), w, dt, units, n, w_idle, "64-clip layers")
    if ring and gemm:
        print("    per launch: recurrent kernel %.1f us, x-projection GEMM %.1f us; recomputed %d"
              % (ring["avg_us"], gemm["avg_us"], sum(m.recompute_count() for m in models)))
    for m in models:
        m.close()


recurrent_loop(None)
if ALT:
    recurrent_loop(ALT)
    os.environ.pop("DSMI_RNN_KERNEL", None)

models = [_native.NativeModel(cfg, sd) for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
feat = torch.from_numpy(syn.make_features(B, T, seed=1)).cuda()
flens = np.full(B, T, dtype=np.int32)


def conv_work(deadline):
    units = 0
    while time.perf_counter() < deadline:
        for k in range(4):
            with torch.cuda.stream(streams[k]):
                models[k].conv_stack(feat, flens)
        units += 4
        if units % 16 == 0:
            torch.cuda.synchronize()
    return units


conv_work(time.perf_counter() + 0.5)
w, dt, units, n, _ = sampled(conv_work, SECS)
report("conv stack", w, dt, units, n, w_idle, "64-clip stacks")
for m in models:
    m.close()

import contextlib, io
from danspeech_amd import Recognizer
from danspeech_amd.deepspeech.model import DeepSpeech
model = DeepSpeech("cfgA", rnn_type="gru", rnn_hidden_size=H, rnn_layers=5, conv_layers=2).load_state_dict(sd)
with contextlib.redirect_stdout(io.StringIO()):
    rec = Recognizer(model=model)
clips = [syn.make_clip(i, 160000) for i in range(32)]
for _ in rec.recognize_batches(clips for _ in range(16)):
    pass


def pipe_work(deadline):
    steps = max(int((deadline - time.perf_counter()) / 0.006), 32)
    for _ in rec.recognize_batches(clips for _ in range(steps)):
        pass
    return steps


w, dt, units, n, _ = sampled(pipe_work, SECS)
report("whole pipeline", w, dt, units, n, w_idle, "32-clip batches")
