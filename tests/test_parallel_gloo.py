"""world_size-2 CPU test (gloo) of the utterance-sharding helpers used by bench.py at N > 1."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from danspeech_amd import parallel
    per, n = 3, 50
    full = np.arange(world * per * n, dtype=np.float64).reshape(world * per, n) if rank == 0 else None
    shard = parallel.scatter_clips(full, per, n, rank, world, torch.device("cpu"))
    exp = np.arange(world * per * n, dtype=np.float64).reshape(world, per, n)[rank]
    ok = np.array_equal(shard.numpy(), exp)
    seqs = [np.arange(rank * 10 + i, rank * 10 + i + (i + rank), dtype=np.int32) for i in range(per)]
    got = parallel.gather_token_ids(seqs, rank, world, torch.device("cpu"))
    if rank == 0:
        want = [np.arange(r * 10 + i, r * 10 + i + (i + r), dtype=np.int32) for r in range(world) for i in range(per)]
        ok = ok and len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))
    else:
        ok = ok and got is None
    # fixed-capacity variant (what bench.py could use to skip the length all-reduce)
    got2 = parallel.gather_token_ids(seqs, rank, world, torch.device("cpu"), cap=8)
    if rank == 0:
        ok = ok and all(np.array_equal(a, b) for a, b in zip(got2, want))
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
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == {0: True, 1: True}
