"""RCCL on the GPU box: a process group of ONE rank (the boxes of this pool have one GPU) running the collectives bench.py and
danspeech_amd/parallel.py issue at N > 1 -- barrier, all_reduce(MAX), gather, scatter, broadcast -- beside a forward of the engine,
in a process of its own.  What it can show: the library loads, a communicator comes up with this image's environment, and its
kernels run next to libdsmi's streams.  What it cannot: anything between two GPUs (tests/test_parallel_gloo.py holds the plan and
the payload formats on two CPU ranks)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
from danspeech_amd import _native, synthetic as syn
cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=256, rnn_layers=2, bidirectional=True, context=20)
m = _native.NativeModel(cfg, syn.make_state_dict(2, "gru", 256, 2, seed=3, **syn.TALKATIVE))
x = torch.from_numpy(syn.make_features(4, 301)).cuda()
lens = np.full(4, 301, dtype=np.int32)
p0, _ = m.forward(x, lens)
torch.cuda.synchronize()
side = torch.cuda.Stream()
for step in range(3):
    p, _ = m.forward(x, lens)                      # the engine's kernels in flight ...
    with torch.cuda.stream(side):                  # ... and the collectives beside them on a stream of their own (bench.py's gather stream)
        t = torch.tensor([1.25 + step], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        g = torch.arange(12, dtype=torch.int32, device=dev).view(3, 4) + step
        outs = [torch.empty_like(g)]
        dist.gather(g, gather_list=outs, dst=0)
        s = torch.empty(5, dtype=torch.int16, device=dev)
        dist.scatter(s, scatter_list=[torch.arange(5, dtype=torch.int16, device=dev) * (step + 1)], src=0)
        b = torch.tensor([7 + step], device=dev)
        dist.broadcast(b, src=0)
    dist.barrier()
    torch.cuda.synchronize()
    assert float(t) == 1.25 + step and torch.equal(outs[0], g) and s.tolist() == [i * (step + 1) for i in range(5)] and int(b) == 7 + step
    assert torch.equal(p, p0)
dist.destroy_process_group()
print("RCCL-ONE-RANK-OK", ".".join(str(v) for v in torch.cuda.nccl.version()))
'''


@pytest.mark.gpu
def test_rccl_collectives_of_one_rank_beside_the_engine():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-ONE-RANK-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.gpu
def test_bench_with_its_process_group_forced_on_one_rank():
    """bench.py's N > 1 plumbing -- init_process_group("nccl", device_id=...), the barriers around the timed region, the MAX
    reduction of the time, the gather stream -- with one rank: the line carries `rccl` and the same parity check as ever."""
    import json
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, DSMI_BENCH_FORCE_GROUP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "4", "--no-side-paths",
                        "--no-other-configs"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["rccl"]["world_size_seen"] == 1 and line["rccl"]["backend"] == "nccl"
    assert line["n_gpus"] == 1 and line["steps"] == 8 and line["value"] > 0
    assert line["parity_checked"] is True


SHARDED = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from danspeech_amd import Recognizer, parallel, synthetic as syn
from danspeech_amd.deepspeech.model import DeepSpeech
sd = syn.make_state_dict(2, "gru", 256, 2, seed=5, **syn.TALKATIVE)
rec = Recognizer(model=DeepSpeech("cfg", rnn_hidden_size=256, rnn_layers=2, conv_layers=2).load_state_dict(sd))
clips = [syn.make_clip(i, 16000 + 1700 * (i %% 7)) for i in range(11)]            # ragged, an odd count, float64 as load_audio gives them
plain = rec.recognize_batch(clips)
parallel._ALWAYS_COLLECTIVE = True            # one rank, but every exchange goes through RCCL
for kind in (np.float64, np.int16, np.float32):
    got = rec.recognize_batch_distributed([np.asarray(c, dtype=kind) for c in clips])
    assert got == plain, (kind, got[:3], plain[:3])
# bench.py's own two exchanges: equal-length int16 clips out, transcripts back
pcm = parallel.scatter_clips(np.stack([np.asarray(syn.make_clip(i, 16000), dtype=np.int16) for i in range(4)]), 4, 16000, 0, 1, torch.device("cuda", 0))
assert pcm.shape == (4, 16000) and pcm.dtype == torch.int16
back = parallel.gather_texts(plain[:4], np.arange(4), 4, 200, 0, 1, torch.device("cuda", 0))
assert back == plain[:4]
ids = parallel.gather_token_ids([np.arange(5, dtype=np.int32), np.arange(2, dtype=np.int32)], 0, 1, torch.device("cuda", 0), 16)
assert [a.tolist() for a in ids] == [[0, 1, 2, 3, 4], [0, 1]]
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL-SHARDED-OK", len(plain), sum(len(t) for t in plain))
'''


@pytest.mark.gpu
def test_the_distributed_entry_end_to_end_over_rccl_on_one_rank():
    """Recognizer.recognize_batch_distributed with every exchange of danspeech_amd/parallel.py forced through the process group
    (header broadcasts, the byte scatter of int16 / float32 / float64 PCM, the failure / width all-reduce, the transcript gather):
    the same calls, payload types and devices N ranks issue, on RCCL, against the plain batch call."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", SHARDED % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RCCL-SHARDED-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
