#!/usr/bin/env python3
"""Headline benchmark: audio-seconds transcribed per wall-second (RTFx) of the
recognize()-equivalent hot path -- spectrogram features -> DeepSpeech forward -> greedy CTC
decode -> strings -- on synthetic 10 s clips, batch 32 per GPU (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  A "step" is one pass of the hot path over one batch of 32 clips per
GPU (weak scaling: per-GPU work is fixed).  Inputs (float64 PCM, what load_audio hands to
recognize(), reference danspeech/audio/resources.py:640) are resident in HBM before the
timed region: at N > 1 rank 0 synthesises all clips and scatters the shards over RCCL; the
per-step result gather (token ids -> rank 0) is inside the timed region.  Weights are seeded
random tensors of the DanSpeech shapes (no network for checkpoints): data = "synthetic".

Rank 0 prints ONE JSON line (see the driver contract), including
  roofline     -- the kernel with the largest total time in the timed region, its algorithmic
                  FLOPs per launch / its mean dispatch duration (per-dispatch begin/end
                  timestamps sampled live through hipExtLaunchKernelGGL events) vs the fp32 MFMA peak
  cpu_baseline -- the numpy oracle (a port of the reference's algorithm) timed on this host
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
# The contraction kernels compute fp32-grade products as three fp16 MFMA products of two-term split
# operands: the ceiling for ALGORITHMIC fp32 FLOPs on that path is the fp16 peak / 3
# (the fp16 and bf16 dense MFMA peaks are the same figure).
PEAK_SPLIT_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0
SPLIT_KERNELS = {"rnn_layer_persistent", "gemm", "gemm_l0", "conv2", "conv3"}
PEAK_HBM_GBS = 8000.0

# HBM/fabric bytes per launch from rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE,
# collected in separate passes (profiles/r01_pmc_*.md); None where no pass has been run.
PMC_TRAFFIC = {"rnn_layer_persistent": 1.24e9 + 0.21e9}   # profiles/r01f_pmc_rnn_persist16.md

CONFIGS = {
    # BASELINE.json configs[1]: "DanSpeechPrimary (5-layer BiRNN, 800 hidden), greedy decode,
    # batch=32 synthetic 10 s 16 kHz clips, 1 MI355X"
    "cfgA-greedy": dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=800, rnn_layers=5, bidirectional=True,
                        context=20, batch=32, seconds=10.0),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfgA-greedy", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default: the config's 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-sampling", action="store_true")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="batches in flight per GPU: each has its own handle set and HIP stream, so the conv/GEMM "
                         "kernels of batch i+1 overlap the latency-bound recurrent chain of batch i")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from danspeech_amd import _native, synthetic as syn
    from danspeech_amd import parallel

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    c = CONFIGS[args.config]
    B = args.batch or c["batch"]
    n_samples = int(c["seconds"] * 16000)
    cfg = {k: c[k] for k in ("conv_layers", "rnn_type", "rnn_hidden_size", "rnn_layers", "bidirectional", "context")}
    labels = syn.DANSPEECH_LABELS
    sd = syn.make_state_dict(cfg["conv_layers"], cfg["rnn_type"], cfg["rnn_hidden_size"], cfg["rnn_layers"],
                             bidirectional=cfg["bidirectional"], seed=0, fc_gain=8.0)
    P = max(1, args.pipeline)
    models = [_native.NativeModel(cfg, sd, device=local, n_labels=len(labels)) for _ in range(P)]
    frontends = [_native.NativeFrontend(device=local) for _ in range(P)]
    decoders = [_native.NativeDecoder(labels, blank_index=0, device=local) for _ in range(P)]
    streams = [torch.cuda.Stream(device=local) for _ in range(P)] if P > 1 else [torch.cuda.current_stream()]
    model = models[0]

    # ---- inputs: rank 0 synthesises, shards go out over RCCL (utterance-level data parallelism)
    if rank == 0:
        all_clips = np.stack([syn.make_clip(i, n_samples) for i in range(B * world)])   # float64 [B*world, N]
    else:
        all_clips = None
    pcm = parallel.scatter_clips(all_clips, B, n_samples, rank, world, torch.device("cuda", local))
    n = np.full(B, n_samples, dtype=np.int64)
    frames = 1 + n // 160
    for mdl in models:
        mdl.reserve(B, int(frames.max()))
    inflight = []          # (context index, probs, out_lens) enqueued but not yet decoded

    def finish(item):
        k, probs, out_lens = item
        with torch.cuda.stream(streams[k]):
            dec = decoders[k].greedy(probs, out_lens)          # synchronises stream k only
        ids = parallel.gather_token_ids([d[0] for d in dec], rank, world, torch.device("cuda", local))
        if rank == 0:
            return ["".join(labels[i] for i in seq) for seq in ids]
        return None

    step_no = [0]

    def step():
        """Enqueue one batch on the next context; retire the oldest batch once P are in flight."""
        k = step_no[0] % P
        step_no[0] += 1
        with torch.cuda.stream(streams[k]):
            feat, fr = frontends[k].features(pcm.view(-1), n)
            probs, out_lens = models[k].forward(feat, fr)
        inflight.append((k, probs, out_lens))
        if len(inflight) >= P:
            return finish(inflight.pop(0))
        return None

    def drain():
        out = None
        while inflight:
            out = finish(inflight.pop(0))
        return out

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(args.warmup):
        out = step()
    out = drain() or out
    if not args.no_kernel_sampling:
        for mdl in models:
            mdl.set_profiling(2)
            mdl.reset_kernel_stats()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step() or out
    out = drain() or out
    sync()
    dt = time.perf_counter() - t0
    stats = {}
    if not args.no_kernel_sampling:
        for mdl in models:                       # merge the per-context samples
            for k, v in mdl.kernel_stats().items():
                a = stats.setdefault(k, dict(launches=0, samples=0, _us=0.0, _fl=0.0, _by=0.0))
                a["launches"] += v["launches"]; a["samples"] += v["samples"]
                a["_us"] += v["avg_us"] * v["samples"]; a["_fl"] += v["flops_per_launch"] * v["launches"]
                a["_by"] += v["bytes_per_launch"] * v["launches"]
            mdl.set_profiling(0)
        for a in stats.values():
            a["avg_us"] = a["_us"] / max(a["samples"], 1)
            a["flops_per_launch"] = a["_fl"] / max(a["launches"], 1)
            a["bytes_per_launch"] = a["_by"] / max(a["launches"], 1)

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    result = None
    if rank == 0:
        audio_s = world * B * c["seconds"] * args.steps
        value = audio_s / dt
        roof = None
        if stats:
            tot = {k: v["avg_us"] * v["launches"] for k, v in stats.items() if v["samples"]}
            dom = max(tot, key=tot.get)
            s = stats[dom]
            ach = s["flops_per_launch"] / (s["avg_us"] * 1e-6) / 1e12
            split = dom in SPLIT_KERNELS and os.environ.get("DSMI_RNN_MODE") != "steps"
            peak = PEAK_SPLIT_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
            roof = dict(bound="mfma", kernel=dom, achieved=round(ach, 3), peak=round(peak, 1), unit="TFLOP/s",
                        frac=round(ach / peak, 4), traffic=PMC_TRAFFIC.get(dom),
                        peak_note=("fp16 dense MFMA peak 2500 TFLOP/s / 3 products per fp32-grade multiply (executed fp16 rate = 3 x achieved)"
                                   if split else "fp32 MFMA peak"),
                        avg_launch_us=round(s["avg_us"], 3), launches_per_step=s["launches"] // args.steps,
                        flops_per_launch=s["flops_per_launch"],
                        kernel_time_share={k: round(v / sum(tot.values()), 4) for k, v in sorted(tot.items())})
        result = {
            "metric": "audio-seconds/sec (RTFx) recognize() on 10 s clips, batch=32",
            "value": round(value, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (two-term fp16 split operands, 3 MFMA products, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: 2conv + 5xBiGRU800 (DanSpeechPrimary per BASELINE), greedy CTC, "
                                   "batch=%d x %.0f s 16 kHz clips per GPU, STFT+forward+decode" % (B, c["seconds"]),
                       "clips_per_gpu": B, "clip_seconds": c["seconds"], "parallelism": "utterance-dp%d" % world,
                       "batches_in_flight": P},
            "roofline": roof,
            # every sampled kernel kind: mean dispatch time and ALGORITHMIC rates (SURVEY 8(d) FLOPs and bytes);
            # conv1/conv2 are the "conv front end" the north star asks GB/s for (HBM spec 8000 GB/s)
            "kernels": {k: {"avg_us": round(v["avg_us"], 2), "launches_per_step": v["launches"] // args.steps,
                            "tflops": round(v["flops_per_launch"] / (v["avg_us"] * 1e-6) / 1e12, 2),
                            "gbps": round(v["bytes_per_launch"] / (v["avg_us"] * 1e-6) / 1e9, 1)}
                        for k, v in sorted(stats.items()) if v["samples"] and v["avg_us"] > 0} if stats else None,
            "sample_transcript_len": len(out[0]) if out else None,
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, sd, B, n_samples, labels)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    for mdl in models:
        mdl.close()
    if world > 1:
        dist.destroy_process_group()
    return result


def cpu_baseline(cfg, sd, B, n_samples, labels):
    """The numpy oracle (kind "port") on this host's cores, one batch of the same workload."""
    from danspeech_amd import synthetic as syn
    from oracle import model as om, features as of, decoder as od
    clips = [syn.make_clip(i, n_samples) for i in range(B)]
    t0 = time.perf_counter()
    feats = np.stack([of.spectrogram(c) for c in clips])[:, None]
    lens = np.full(B, feats.shape[-1])
    probs, out_lens = om.forward(sd, cfg, feats, lens)
    od.greedy_decode(probs, out_lens, labels, 0)
    dt = time.perf_counter() - t0
    return {"value": round(B * n_samples / 16000.0 / dt, 2), "unit": "audio-s/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "one batch of %d x %.0f s clips through oracle/ (numpy fp32, BLAS threads = all cores): "
                      "features + forward + greedy, %.1f s wall" % (B, n_samples / 16000.0, dt)}


if __name__ == "__main__":
    main()
