#!/usr/bin/env python3
"""Hunting the four-wave ring kernel's small-shape defect (profiles/r06_ring4_small_shapes.txt) below the Python engine: P model handles
of one small GRU on P streams, the same 64-clip batch on each, forwards enqueued together; every output against the same handle's
output when it ran alone.  Prints, per round with a mismatch: handle, clips (position in the batch = 16 * tile + clip), first wrong
output frame, direction of the damage in time, NaNs.
    DSMI_RNN_KERNEL=ring4 python tools/exp/ring4_race.py [H=128] [handles=4] [rounds=12] [equal|ragged]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from danspeech_amd import _native, synthetic as syn
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
R = int(sys.argv[3]) if len(sys.argv) > 3 else 12
equal = len(sys.argv) > 4 and sys.argv[4] == "equal"
cfg = dict(conv_layers=2, rnn_type=os.environ.get("DBG_KIND", "gru"), rnn_hidden_size=H, rnn_layers=int(os.environ.get("DBG_LAYERS", "3")), bidirectional=True, context=20)
sd = syn.make_state_dict(2, cfg["rnn_type"], H, cfg["rnn_layers"], seed=12, fc_gain=8.0)
NB = int(os.environ.get("DBG_B", "64"))
L0 = int(os.environ.get("DBG_LEN", "9000"))     # samples of the shortest clip (9000: ~30 output frames)
clips = ([syn.make_clip(i, L0 + 800 if equal else L0 + 400 * (i % 5)) for i in range(32)] * 2)[:NB]
order = np.argsort([-len(c) for c in clips], kind="stable")
n = np.array([len(clips[i]) for i in order], dtype=np.int64)
fe = _native.NativeFrontend()
feat, frames = fe.features(torch.from_numpy(np.concatenate([clips[i] for i in order])).cuda(), n)
torch.cuda.synchronize()
models = [_native.NativeModel(cfg, sd) for _ in range(P)]
streams = [torch.cuda.Stream() for _ in range(P)]
for m in models:
    m.set_inflight(4)
ref = []
for m in models:                                  # alone on the chip, one after the other
    p, ol = m.forward(feat, frames)
    torch.cuda.synchronize()
    ref.append(p.clone())
ol = np.asarray(ol)
print("H %d, %d handles, %s lengths, out_lens %d..%d; alone: handles agree with handle 0 to %.2e" %
      (H, P, "equal" if equal else "ragged", ol.min(), ol.max(), max(float((r - ref[0]).abs().max()) for r in ref)), flush=True)
bad_rounds = 0
burn = os.environ.get("DBG_BURN")
if burn:
    bs = torch.cuda.Stream()
    ba = torch.randn(4096, 4096, device="cuda"); bb = torch.randn(4096, 4096, device="cuda")
for rnd in range(R):
    outs = []
    if burn:
        with torch.cuda.stream(bs):
            for _ in range(int(burn)):
                bc = ba @ bb
    for m, s in zip(models, streams):
        with torch.cuda.stream(s):
            p, _ = m.forward(feat, frames, check=False)
            outs.append(p)
    torch.cuda.synchronize()
    for m in models:
        m.status()
    for k, (p, r) in enumerate(zip(outs, ref)):
        d = (p - r).abs()
        nan = int(torch.isnan(p).sum())
        d = torch.nan_to_num(d, nan=9.0).cpu().numpy()
        per_clip = np.array([d[b, :ol[b]].max() for b in range(NB)])
        wrong = np.nonzero(per_clip > 1e-4)[0]
        if len(wrong):
            bad_rounds += 1
            b = int(wrong[0])
            fr = np.nonzero(d[b, :ol[b]].max(-1) > 1e-4)[0]
            print("round %2d handle %d: %2d clips wrong %s; clip %d (tile %d, %d frames): wrong frames %d..%d (%d of them), max err %.3g, NaNs %d"
                  % (rnd, k, len(wrong), wrong.tolist(), b, b // 16, ol[b], fr.min(), fr.max(), len(fr), per_clip.max(), nan), flush=True)
print("rounds with a mismatch: %d of %d x %d; recomputed %s" % (bad_rounds, R, P, [m.recompute_count() for m in models]))
