import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from danspeech_amd import synthetic as syn, _native
sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, fc_gain=8.0)
cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=800, rnn_layers=5, bidirectional=True, context=20)
m = _native.NativeModel(cfg, sd)
for B in (16, 32, 64, 96, 128):
    x = torch.from_numpy(syn.make_features(B, 1001)).cuda()
    lens = np.full(B, 1001, dtype=np.int32)
    m.forward(x, lens); torch.cuda.synchronize()
    m.set_profiling(2); m.reset_kernel_stats()
    for _ in range(3): m.forward(x, lens)
    torch.cuda.synchronize()
    ks = m.kernel_stats()
    print("B=%3d persist %.0f us/layer = %.2f us/step (%.2f us per 32-clip tile-step)  gemm %.0f conv2 %.0f" % (
        B, ks["rnn_layer_persistent"]["avg_us"], ks["rnn_layer_persistent"]["avg_us"] / 501,
        ks["rnn_layer_persistent"]["avg_us"] / 501 / max(1, (B + 31) // 32), ks["gemm"]["avg_us"], ks["conv2"]["avg_us"]), flush=True)
    m.set_profiling(0)
