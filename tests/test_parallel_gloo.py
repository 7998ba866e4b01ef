"""world_size-2 CPU test (gloo) of the utterance-sharding helpers used by bench.py at N > 1."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _EchoEngine:
    """Stands in for DanSpeechRecognizer on the CPU: the "transcript" of a clip is a function of its samples, so the
    test can tell that every clip reached exactly one rank intact and came back at its own position."""

    fail_on = None

    def transcribe_device(self, pcm, n_samples, show_all=False, max_batch=None):
        out, off = [], 0
        if self.fail_on is not None and dist.get_rank() == self.fail_on:
            raise ValueError("boom on rank %d" % self.fail_on)
        assert list(n_samples) == sorted(n_samples, reverse=True)          # each shard arrives longest first
        for n in n_samples:
            clip = pcm[off:off + int(n)].to(torch.float64).numpy()
            off += int(n)
            text = "n%d s%d \u00e6\u00f8" % (len(clip), int(clip.sum()))
            out.append([text, text + "!", "x" * (len(clip) % 7)] if show_all else text)
        assert off == pcm.numel()
        return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from danspeech_amd import parallel
    cpu = torch.device("cpu")
    per, n = 3, 50
    full = (np.arange(world * per * n) % 3000 - 1500).astype(np.int16).reshape(world * per, n)
    shard = parallel.scatter_clips(full if rank == 0 else None, per, n, rank, world, cpu)
    ok = shard.dtype == torch.int16 and np.array_equal(shard.numpy(), full.reshape(world, per, n)[rank])
    seqs = [np.arange(rank * 10 + i, rank * 10 + i + (i + rank), dtype=np.int32) for i in range(per)]
    got = parallel.gather_token_ids(seqs, rank, world, cpu, cap=8)
    if rank == 0:
        want = [np.arange(r * 10 + i, r * 10 + i + (i + r), dtype=np.int32) for r in range(world) for i in range(per)]
        ok = ok and len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))
    else:
        ok = ok and got is None
    # end to end: ragged clips in arbitrary order, odd count (ranks get 4 and 3), results back in the caller's order
    for dtype in (np.int16, np.float64):
        rng = np.random.default_rng(5)
        lens = [700, 160, 1601, 333, 1601, 20, 999]
        clips = [rng.integers(-3000, 3000, size=k).astype(dtype) for k in lens]
        res = parallel.recognize_sharded(_EchoEngine(), clips if rank == 0 else None, rank, world, cpu, frames_cap=64)
        if rank == 0:
            ok = ok and res == ["n%d s%d \u00e6\u00f8" % (len(c), int(c.astype(np.float64).sum())) for c in clips]
        else:
            ok = ok and res is None
    # all beams of every clip (lists of strings, padded to the longest transcript of any rank)
    rng = np.random.default_rng(6)
    clips = [rng.integers(-3000, 3000, size=k).astype(np.int16) for k in (90, 401, 33, 250, 12)]
    res = parallel.recognize_sharded(_EchoEngine(), clips if rank == 0 else None, rank, world, cpu, show_all=True)
    if rank == 0:
        base = ["n%d s%d \u00e6\u00f8" % (len(c), int(c.astype(np.float64).sum())) for c in clips]
        ok = ok and res == [[b, b + "!", "x" * (len(c) % 7)] for b, c in zip(base, clips)]
    else:
        ok = ok and res is None
    # an engine that fails on ONE rank: every rank raises, nobody is left waiting in the gather
    eng = _EchoEngine()
    eng.fail_on = 1
    try:
        parallel.recognize_sharded(eng, clips if rank == 0 else None, rank, world, cpu)
        ok = False
    except ValueError as e:
        ok = ok and rank == 1 and "boom" in str(e)
    except RuntimeError as e:
        ok = ok and rank == 0 and "another rank failed" in str(e)
    # fewer clips than ranks: one rank gets nothing
    res = parallel.recognize_sharded(_EchoEngine(), [np.ones(40, np.float32)] if rank == 0 else None, rank, world, cpu, frames_cap=64)
    ok = ok and (res == ["n40 s40 \u00e6\u00f8"] if rank == 0 else res is None)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_scatter_gather_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=90) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == {0: True, 1: True}
