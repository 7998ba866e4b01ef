"""Per-kernel dispatch times of one forward at a time (one batch in flight, nothing beside it): the clean figures of DESIGN §4."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from danspeech_amd import _native, synthetic as syn
cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=800, rnn_layers=5, bidirectional=True, context=20)
sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, **syn.TALKATIVE)
m = _native.NativeModel(cfg, sd)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.from_numpy(syn.make_features(B, 1001)).cuda()
lens = np.full(B, 1001, dtype=np.int32)
for _ in range(3):
    m.forward(x, lens)
m.set_profiling(2); m.reset_kernel_stats()
for _ in range(10):
    m.forward(x, lens)
torch.cuda.synchronize()
for k, v in sorted(m.kernel_stats().items()):
    if v["samples"]:
        print("%-22s %8.1f us  %7.1f TF-equiv" % (k, v["avg_us"], v["flops_per_launch"] / (v["avg_us"] * 1e-6) / 1e12))
