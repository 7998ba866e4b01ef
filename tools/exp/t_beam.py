import os, sys, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
src = open(os.path.join(ROOT, "tools", "run_configs.py")).read().split("which = [")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "run_configs.py")}; exec(compile(src, "rc", "exec"), ns)
from danspeech_amd import synthetic as syn
rec = ns["build"](800, 5, 3, 64)
eng = rec.danspeech_recognizer
clips = [syn.make_clip(i, 160000) for i in range(32)]
for _ in range(5): rec.recognize_batch(clips)
order = np.argsort([-len(r) for r in clips], kind="stable")
feats, frames = eng.audio_parser.parse_batch([clips[j] for j in order])
probs, sizes = eng.model.enqueue(feats, torch.from_numpy(frames.astype(np.int32))); eng.model.collect(); torch.cuda.synchronize()
dec = eng.decoder._dec(0)
sz = np.asarray(torch.as_tensor(sizes).cpu()).astype(np.int32)
for i in range(3):
    t0 = time.perf_counter(); out = dec.beam(probs, sz, beam_width=64, cutoff_top_n=40, cutoff_prob=1.0); t1 = time.perf_counter()
    s, o = eng.decoder.decode(probs, sizes); t2 = time.perf_counter()
    print("native beam call %.1f ms, full decode %.1f ms -> python part %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t2-t1-(t1-t0))*1e3))
