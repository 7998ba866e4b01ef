import time, sys
sys.path.insert(0, '.')
from danspeech_amd import _native, synthetic as syn
t=time.time(); sd = syn.make_state_dict(2, "gru", 800, 5, seed=0); print("sd", time.time()-t)
cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=800, rnn_layers=5, bidirectional=True, context=20)
t=time.time(); m=_native.NativeModel(cfg, sd); print("model", time.time()-t)
