"""Utterance-level data parallelism over the GPUs of one node.

The reference has no distributed code at all (no NCCL/MPI/Gloo call sites); clips
are independent on this path (per-clip feature normalisation, reference
danspeech/audio/parsers.py:66-70; eval-mode BatchNorm; MaskConv/packing isolate sequences,
danspeech/deepspeech/model.py:57-58,117), so the only exchanges are the input scatter and the
result gather.  One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests); weights are replicated.
"""
import numpy as np


def plan_shards(lengths, world):
    """Sort clips by length (descending, stable) and deal them round-robin so every rank gets a
    similar length mix and a locally descending order (pack_padded_sequence's requirement,
    model.py:117).  Returns a list of index arrays, one per rank."""
    order = np.argsort(-np.asarray(lengths, dtype=np.int64), kind="stable")
    return [order[r::world] for r in range(world)]


def scatter_clips(all_clips, per_rank, n_samples, rank, world, device):
    """rank 0 holds ``all_clips`` float64 [world*per_rank, n_samples]; every rank returns its
    [per_rank, n_samples] shard on ``device``."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return torch.from_numpy(np.ascontiguousarray(all_clips)).to(device)
    out = torch.empty((per_rank, n_samples), dtype=torch.float64, device=device)
    chunks = None
    if rank == 0:
        full = torch.from_numpy(np.ascontiguousarray(all_clips)).to(device).view(world, per_rank, n_samples)
        chunks = [full[r].contiguous() for r in range(world)]
    dist.scatter(out, scatter_list=chunks, src=0)
    return out


def gather_token_ids(seqs, rank, world, device, cap=None):
    """Gather per-utterance int32 token-id arrays to rank 0 (rank-major order).  Fixed-size
    padded payload: column 0 is the length.  Returns the list on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return [np.asarray(s) for s in seqs]
    if cap is None:
        mx = torch.tensor([max([len(s) for s in seqs] + [0])], dtype=torch.int64, device=device)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        cap = int(mx.item())
    buf = np.zeros((len(seqs), cap + 1), dtype=np.int32)
    for i, s in enumerate(seqs):
        buf[i, 0] = len(s)
        buf[i, 1:1 + len(s)] = s
    t = torch.from_numpy(buf).to(device)
    outs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, gather_list=outs, dst=0)
    if rank != 0:
        return None
    res = []
    for o in outs:
        a = o.cpu().numpy()
        res.extend(a[i, 1:1 + a[i, 0]].copy() for i in range(a.shape[0]))
    return res
