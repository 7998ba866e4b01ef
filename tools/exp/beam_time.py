import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from danspeech_amd import synthetic as syn
from danspeech_amd.deepspeech.model import DeepSpeech
from danspeech_amd.deepspeech.decoder import BeamCTCDecoder

H, L = 800, 5
sd = syn.make_state_dict(2, "gru", H, L, seed=0, fc_gain=8.0)
m = DeepSpeech("cfg", rnn_hidden_size=H, rnn_layers=L).load_state_dict(sd).to("cuda")
path = os.path.join(tempfile.gettempdir(), "syn3.arpa")
syn.make_arpa(path, order=3, n_words=5000, seed=11, ngrams_per_order=20000)
B = 32
x = torch.from_numpy(syn.make_features(B, 1001)).cuda()
lens = torch.full((B,), 1001, dtype=torch.int32)
probs, sizes = m(x, lens)
for beam, lm in ((64, None), (64, path), (128, path)):
    dec = BeamCTCDecoder(syn.DANSPEECH_LABELS, lm_path=lm, alpha=1.3, beta=0.2, beam_width=beam)
    dec.decode(probs, sizes)
    nd = dec._dec(0)
    sz = np.asarray(sizes.cpu()).astype(np.int32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): nd.beam(probs, sz, beam_width=beam)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(3): out = dec.decode(probs, sizes)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("beam=%d lm=%s: native %.1f ms, decode() %.1f ms; top: %r" % (beam, bool(lm), (t1 - t0) / 3e-3, (t2 - t1) / 3e-3, out[0][0][0][:60]), flush=True)
