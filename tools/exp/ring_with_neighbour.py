#!/usr/bin/env python3
"""One recurrent layer (ring kernel, 64 clips, cfgA's width) beside a neighbour that (0) does nothing, (1) multiplies in registers on
the other CUs (power, no memory traffic), (2) streams HBM on the other CUs: where does the ring kernel's 24 % in the pipeline come
from -- the clock or the memory system?   ring_with_neighbour.py [workgroups of the neighbour = 412]"""
import os, sys, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from danspeech_amd import _native, synthetic as syn
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libburner.so"))
lib.burn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
WGS = int(sys.argv[1]) if len(sys.argv) > 1 else 412          # 206 CUs x 2 workgroups of 256 threads: what the ring's 50 CUs leave
H, B, T = 800, 64, 1001
cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=H, rnn_layers=2, bidirectional=True, context=20)
m = _native.NativeModel(cfg, syn.make_state_dict(2, "gru", H, 2, seed=0))
m.set_inflight(2)
m.set_profiling(2)
x = torch.from_numpy(syn.make_features(B, T, seed=1)).cuda()
lens = np.full(B, T, dtype=np.int32)
buf = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
sink = torch.zeros(256, dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()
for _ in range(3):
    m.forward(x, lens)
torch.cuda.synchronize()
def ring_us():
    v = m.kernel_stats()["rnn_layer_persistent"]
    return v["avg_us"], v["launches"]
ga = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
gb = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
for name, mode, iters in (("alone", -1, 0), ("MFMA burner on the other CUs", 0, 60000), ("HBM streamer on the other CUs", 1, 24),
                          ("fp16 GEMMs (8192^3, library) beside it", 2, 40), ("alone again", -1, 0)):
    a0, n0 = ring_us()
    if mode >= 0:
        t0 = time.perf_counter()
        if mode == 2:
            with torch.cuda.stream(side):
                for _ in range(iters):
                    gc = ga @ gb
        else:
            lib.burn(mode, WGS, iters, buf.data_ptr(), buf.numel(), sink.data_ptr(), side.cuda_stream)
        time.sleep(0.002)                                       # the neighbour is running when the forwards start
    for _ in range(2):
        m.forward(x, lens)
    torch.cuda.current_stream().synchronize()
    busy = None
    if mode >= 0:
        still = not side.query()
        side.synchronize()
        busy = (time.perf_counter() - t0) * 1e3
    a1, n1 = ring_us()
    per = (a1 * n1 - a0 * n0) / max(n1 - n0, 1)
    print("%-32s ring launch %7.1f us (%d launches)%s" % (name, per, n1 - n0,
          "" if busy is None else "; neighbour ran %.1f ms, %s when the forwards ended" % (busy, "still running" if still else "ALREADY DONE: lengthen it")), flush=True)
