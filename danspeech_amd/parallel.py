"""Utterance-level data parallelism over the GPUs of one node.

The reference has no distributed code at all (no NCCL/MPI/Gloo call sites); clips
are independent on this path (per-clip feature normalisation, reference
danspeech/audio/parsers.py:66-70; eval-mode BatchNorm; MaskConv/packing isolate sequences,
danspeech/deepspeech/model.py:57-58,117), so the only exchanges are the input scatter and the
result gather.  One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests); weights are replicated; no collective inside the model.

Payloads are kept small and fixed-size: PCM travels in the clips' own sample type (int16 for
audio files: 320 KB per 10 s clip, not the 1.28 MB of float64), results as int32 token ids
padded to a cap every rank can compute from the clip lengths (no per-step reduction to agree
on a width).
"""
import numpy as np

_PCM_CODES = {np.dtype(np.int16): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}
_PCM_TYPES = {v: k for k, v in _PCM_CODES.items()}


def plan_shards(lengths, world):
    """Sort clips by length (descending, stable) and deal them round-robin so every rank gets a
    similar length mix and a locally descending order (pack_padded_sequence's requirement,
    model.py:117).  Returns a list of index arrays, one per rank."""
    order = np.argsort(-np.asarray(lengths, dtype=np.int64), kind="stable")
    return [order[r::world] for r in range(world)]


# tests/test_gpu_rccl_one_rank.py: with a process group of ONE rank every function here still goes through its collectives
# (a one-GPU box can then run the same calls, payload types and devices over RCCL that N ranks would)
_ALWAYS_COLLECTIVE = False


def _alone(world):
    return world == 1 and not _ALWAYS_COLLECTIVE


def _torch_dtype(np_dtype):
    import torch
    return {np.dtype(np.int16): torch.int16, np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[np.dtype(np_dtype)]


def _bytes(t):
    """The tensor's storage as uint8: RCCL / gloo have no 16-bit integer type, and a scatter moves bytes anyway."""
    import torch
    return t.view(torch.uint8)


def scatter_clips(all_clips, per_rank, n_samples, rank, world, device, dtype=np.int16):
    """Equal-length clips (bench.py): rank 0 holds ``all_clips`` [world*per_rank, n_samples] of ``dtype``;
    every rank returns its [per_rank, n_samples] shard on ``device``."""
    import torch
    import torch.distributed as dist
    if _alone(world):
        return torch.from_numpy(np.ascontiguousarray(all_clips, dtype=dtype)).to(device)
    out = torch.empty((per_rank, n_samples), dtype=_torch_dtype(dtype), device=device)
    chunks = None
    if rank == 0:
        full = torch.from_numpy(np.ascontiguousarray(all_clips, dtype=dtype)).to(device).view(world, per_rank, n_samples)
        chunks = [_bytes(full[r].contiguous()) for r in range(world)]
    dist.scatter(_bytes(out), scatter_list=chunks, src=0)
    return out


def gather_token_ids(seqs, rank, world, device, cap):
    """Gather per-utterance int32 token-id arrays to rank 0 (rank-major order).  Fixed-size padded payload of
    ``cap`` ids per utterance (a CTC transcript is never longer than the frame count, so ``cap`` = output frames
    is known to every rank); column 0 is the length.  Returns the list on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    if _alone(world):
        return [np.asarray(s) for s in seqs]
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


def scatter_ragged(clips, rank, world, device):
    """Rank 0 passes a list of 1-D clips (one sample type); every rank gets back
    ``(pcm, n_samples, indices)``: its shard's clips back to back on ``device`` (longest first), their lengths, and
    their positions in the caller's list.  Two collectives: a broadcast of the header (count, sample type, lengths)
    and one scatter of equal-size padded payloads."""
    import torch
    import torch.distributed as dist
    head = torch.zeros(2, dtype=torch.int64, device=device)
    if rank == 0:
        kinds = {np.asarray(c).dtype for c in clips}
        dtype = kinds.pop() if len(kinds) == 1 and next(iter(kinds)) in _PCM_CODES else np.dtype(np.float64)
        head[0], head[1] = len(clips), _PCM_CODES[dtype]
    if not _alone(world):
        dist.broadcast(head, src=0)
    count, dtype = int(head[0]), _PCM_TYPES[int(head[1])]
    lens = torch.zeros(max(count, 1), dtype=torch.int64, device=device)
    if rank == 0 and count:
        lens[:count] = torch.tensor([len(c) for c in clips], dtype=torch.int64)
    if not _alone(world):
        dist.broadcast(lens, src=0)
    lengths = lens[:count].cpu().numpy()
    shards = plan_shards(lengths, world)
    totals = [int(lengths[s].sum()) for s in shards]
    width = max(max(totals), 1)
    mine = torch.empty(width, dtype=_torch_dtype(dtype), device=device)
    if _alone(world):
        mine[:totals[0]] = torch.from_numpy(np.concatenate([np.asarray(clips[i], dtype=dtype) for i in shards[0]])) if count else mine[:0]
    else:
        payload = None
        if rank == 0:
            host = np.zeros((world, width), dtype=dtype)
            for r, s in enumerate(shards):
                if len(s):
                    host[r, :totals[r]] = np.concatenate([np.asarray(clips[i], dtype=dtype) for i in s])
            full = torch.from_numpy(host).to(device)
            payload = [_bytes(full[r].contiguous()) for r in range(world)]
        dist.scatter(_bytes(mine), scatter_list=payload, src=0)
    idx = shards[rank]
    return mine[:totals[rank]], lengths[idx].astype(np.int64), idx, lengths


def gather_texts(texts, indices, count, cap, rank, world, device, beams=1):
    """Every rank passes the transcripts of its shard (``indices`` = their positions in the original list); rank 0
    returns the ``count`` transcripts in the caller's order.  Fixed payload: ``cap`` UTF-32 code points + length +
    position per clip, one ``gather``.  With ``beams`` > 1 every element of ``texts`` is a list of up to that many
    strings (all beams of a clip, best first) and lists come back."""
    import torch
    import torch.distributed as dist
    per = (count + world - 1) // world                       # plan_shards gives every rank at most this many
    buf = np.zeros((per, beams, cap + 2), dtype=np.int32)
    buf[:, :, 0] = -1
    for row, (text, i) in enumerate(zip(texts, indices)):
        for k, one in enumerate([text] if beams == 1 else list(text)[:beams]):
            codes = np.frombuffer(one.encode("utf-32-le"), dtype="<u4").astype(np.int32)
            if len(codes) > cap:
                raise ValueError("transcript longer than the frame count it was decoded from")
            buf[row, k, 0], buf[row, k, 1] = i, len(codes)
            buf[row, k, 2:2 + len(codes)] = codes
    t = torch.from_numpy(buf).to(device)
    if _alone(world):
        outs = [t]
    else:
        outs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, gather_list=outs, dst=0)
        if rank != 0:
            return None
    res = [None] * count if beams == 1 else [[] for _ in range(count)]
    for o in outs:
        a = o.cpu().numpy()
        for clip in a:
            for row in clip:
                if row[0] < 0:
                    continue
                text = row[2:2 + row[1]].astype("<u4").tobytes().decode("utf-32-le")
                if beams == 1:
                    res[int(row[0])] = text
                else:
                    res[int(row[0])].append(text)
    return res


def recognize_sharded(engine, clips, rank, world, device, frames_cap=None, show_all=False, max_batch=32):
    """``engine.transcribe_batch(clips, show_all)`` over ``world`` ranks: rank 0 passes the list and gets the results
    back in its order; the other ranks pass ``None`` and get ``None``.  ``engine`` needs
    ``transcribe_device(pcm, n_samples, show_all, max_batch)`` (``DanSpeechRecognizer``: the shard runs as a pipelined
    sequence of batches of at most ``max_batch`` clips; the CPU tests pass a stand-in).  Exchanges: the header broadcasts
    and ONE scatter in, ONE gather out, and between them one three-word all-reduce (MAX): "some rank failed" (then every
    rank raises instead of one rank leaving the others in the gather), the beam count and the longest transcript (the
    width every rank pads to when all beams travel)."""
    import torch
    import torch.distributed as dist
    pcm, n, idx, lengths = scatter_ragged(clips, rank, world, device)
    # every validation that can fail on one rank only comes BEFORE the gather, and a failure is shared: a rank that raised
    # alone would leave the others waiting in the collective for ever
    err, results = None, []
    try:
        results = engine.transcribe_device(pcm, n, show_all=show_all, max_batch=max_batch) if len(n) else []
    except Exception as e:           # noqa: BLE001 -- re-raised below on every rank
        err = e
    cap = frames_cap if frames_cap is not None else int(lengths.max() // 160 + 1) if len(lengths) else 1
    beams = 1
    longest = max([len(t) for r in results for t in ([r] if not show_all else r)] + [0])
    if not _alone(world):
        word = torch.tensor([1 if err is not None else 0, max([len(r) for r in results] + [1]) if show_all else 1, longest],
                            dtype=torch.int64, device=device)
        dist.all_reduce(word, op=dist.ReduceOp.MAX)
        failed, beams, longest = int(word[0]), int(word[1]), int(word[2])
    else:
        failed, beams = int(err is not None), (max([len(r) for r in results] + [1]) if show_all else 1)
    if failed:
        raise err if err is not None else RuntimeError("recognize_sharded: another rank failed")
    if show_all:
        cap = max(longest, 1)                      # beams x frames code points per clip would be megabytes: use the real width
    return gather_texts(results, idx, len(lengths), cap, rank, world, device, beams=beams)
