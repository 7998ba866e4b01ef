import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from danspeech_amd import synthetic as syn, _native
for H in (112, 208, 400, 608, 800, 1008):
    sd = syn.make_state_dict(2, "gru", H, 2, seed=0, fc_gain=8.0)
    cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=H, rnn_layers=2, bidirectional=True, context=20)
    m = _native.NativeModel(cfg, sd)
    B = 32
    x = torch.from_numpy(syn.make_features(B, 1001)).cuda()
    lens = np.full(B, 1001, dtype=np.int32)
    m.forward(x, lens); torch.cuda.synchronize()
    m.set_profiling(2); m.reset_kernel_stats()
    for _ in range(3): m.forward(x, lens)
    torch.cuda.synchronize()
    ks = m.kernel_stats()
    print("H=%4d (%3d workgroups/direction, %3d KB of state per workgroup per step): %.2f us/step" % (
        H, H // 16, H * 16 * 4 // 1024, ks["rnn_layer_persistent"]["avg_us"] / 501), flush=True)
    m.close()
