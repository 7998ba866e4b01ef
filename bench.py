#!/usr/bin/env python3
"""Headline benchmark: audio-seconds transcribed per wall-second (RTFx) of the
recognize()-equivalent hot path -- spectrogram features -> DeepSpeech forward -> greedy CTC
decode -> strings -- on synthetic 10 s clips, batch 32 per GPU (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  A "step" is one pass of the hot path over one batch of 32 clips per
GPU (weak scaling: per-GPU work is fixed).  Inputs (float64 PCM, what load_audio hands to
recognize(), reference danspeech/audio/resources.py:640) are resident in HBM before the
timed region: at N > 1 rank 0 synthesises all clips and scatters the shards over RCCL as
int16 (the clips' on-disk type; widened to float64 on the device before the timed region);
the per-step result gather (token ids -> rank 0, fixed-size payload) is inside the timed
region.  Weights are seeded random tensors of the DanSpeech shapes (no network for
checkpoints): data = "synthetic"; they are scaled so that a 10 s clip decodes to 100+
tokens (synthetic.TALKATIVE), and the timed batch is CHECKED against the CPU oracle.

Rank 0 prints ONE JSON line (see the driver contract), including
  roofline       -- the kernel with the largest total time in the timed region: algorithmic FLOPs per
                    launch / its mean dispatch duration (per-dispatch begin/end timestamps sampled live
                    through hipExtLaunchKernelGGL events) vs the MFMA peak of the instruction it runs on
  cpu_baseline   -- oracle/torch_port.py (the reference's own CPU operators: oneDNN conv, aten::gru, ...;
                    kind "port") on this host's physical cores, same batch
  parity         -- max |probs - oracle| and transcript equality of the GPU's batch vs that oracle run
  public_surface -- the same workload through Recognizer.recognize_batches (host float64 arrays in,
                    strings out: staging + PCIe upload included), N = 1 only
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
SPLIT_KERNELS = {"rnn_layer_persistent", "gemm", "gemm_l0", "conv1", "conv2", "conv3"}
PEAK_HBM_GBS = 8000.0

# HBM/fabric bytes per launch from rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, collected in
# separate passes on this workload (profiles/r02_pmc_*.md, tools/pmc_summary.py); None where no pass has been run.
PMC_TRAFFIC = {}
try:
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as _f:
        PMC_TRAFFIC = json.load(_f)
except OSError:
    pass

CONFIGS = {
    # BASELINE.json configs[1]: "DanSpeechPrimary (5-layer BiRNN, 800 hidden), greedy decode,
    # batch=32 synthetic 10 s 16 kHz clips, 1 MI355X"
    "cfgA-greedy": dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=800, rnn_layers=5, bidirectional=True,
                        context=20, batch=32, seconds=10.0),
}


def physical_cores():
    """(count, model name) of this host's physical cores from /proc/cpuinfo."""
    cores, model, phys, core = set(), "unknown", None, None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    n = len(cores) or (os.cpu_count() or 1)
    return min(n, os.cpu_count() or n), model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfgA-greedy", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default: the config's 32)")
    ap.add_argument("--hidden", type=int, default=None, help="experiments: another hidden size (the JSON line then names it)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle run (and with it the parity check)")
    ap.add_argument("--no-public-surface", action="store_true")
    ap.add_argument("--no-kernel-sampling", action="store_true")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="batches in flight per GPU (default 2): each has its own handle set and HIP stream; the recurrent "
                         "layers of the two batches share every CU (half-CU persistent workgroups, one gate lane each) and "
                         "the conv/GEMM kernels of one batch run in the waits of the other's recurrent chain")
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
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    c = dict(CONFIGS[args.config])
    if args.hidden:
        c["rnn_hidden_size"] = args.hidden
    B = args.batch or c["batch"]
    n_samples = int(c["seconds"] * 16000)
    cfg = {k: c[k] for k in ("conv_layers", "rnn_type", "rnn_hidden_size", "rnn_layers", "bidirectional", "context")}
    labels = syn.DANSPEECH_LABELS
    sd = syn.make_state_dict(cfg["conv_layers"], cfg["rnn_type"], cfg["rnn_hidden_size"], cfg["rnn_layers"],
                             bidirectional=cfg["bidirectional"], seed=0, **syn.TALKATIVE)
    P = max(1, args.pipeline)
    models = [_native.NativeModel(cfg, sd, device=local, n_labels=len(labels)) for _ in range(P)]
    for mdl in models:
        mdl.set_inflight(P)
    frontends = [_native.NativeFrontend(device=local) for _ in range(P)]
    decoders = [_native.NativeDecoder(labels, blank_index=0, device=local) for _ in range(P)]
    streams = [torch.cuda.Stream(device=local) for _ in range(P)] if P > 1 else [torch.cuda.current_stream()]
    model = models[0]

    # ---- inputs: rank 0 synthesises, shards go out over RCCL as int16 (utterance-level data parallelism)
    if rank == 0:
        all_clips = np.stack([syn.make_clip(i, n_samples) for i in range(B * world)])   # integer-valued float64 [B*world, N]
    else:
        all_clips = None
    pcm = parallel.scatter_clips(all_clips, B, n_samples, rank, world, dev, dtype=np.int16).to(torch.float64)
    n = np.full(B, n_samples, dtype=np.int64)
    frames = 1 + n // 160
    for mdl in models:
        mdl.reserve(B, int(frames.max()))
    To = int(model.seq_lens(np.array([int(frames.max())], dtype=np.int32))[0])
    inflight = []          # (context index, probs, out_lens) enqueued but not yet decoded
    last = {}

    def finish(item):
        k, probs, out_lens = item
        models[k].status()                                      # a timed-out batch is recomputed here, never decoded as garbage
        with torch.cuda.stream(streams[k]):
            dec = decoders[k].greedy(probs, out_lens)          # synchronises stream k only
        last["probs"], last["out_lens"] = probs, out_lens
        ids = parallel.gather_token_ids([d[0] for d in dec], rank, world, dev, cap=To)
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
            probs, out_lens = models[k].forward(feat, fr, check=False)
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
    recomputed = sum(mdl.recompute_count() for mdl in models)

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
                        frac=round(ach / peak, 4), traffic=(PMC_TRAFFIC.get(dom) or {}).get("bytes_per_launch"),
                        peak_note=("fp16 dense MFMA peak 2500 TFLOP/s / 3 products per fp32-grade multiply (executed fp16 rate = 3 x achieved)"
                                   if split else "fp32 MFMA peak"),
                        # with P batches in flight P launches of the recurrent kernel run at a time, each on its own lane of CUs:
                        # `achieved` / `frac` are per launch (the contract's definition); the rate the chip sustains while they
                        # run is `concurrent_launches` times that
                        concurrent_launches=(min(P, 2) if dom == "rnn_layer_persistent" else 1),
                        frac_all_concurrent_launches=round(ach / peak * (min(P, 2) if dom == "rnn_layer_persistent" else 1), 4),
                        avg_launch_us=round(s["avg_us"], 3), launches_per_step=s["launches"] // args.steps,
                        flops_per_launch=s["flops_per_launch"],
                        kernel_time_share={k: round(v / sum(tot.values()), 4) for k, v in sorted(tot.items())})
        result = {
            "metric": "audio-seconds/sec (RTFx) recognize() on 10 s clips, batch=32",
            "value": round(value, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (two-term fp16 split operands, 3 MFMA products, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: 2conv + 5xBiGRU%d (DanSpeechPrimary per BASELINE), greedy CTC, "
                                   "batch=%d x %.0f s 16 kHz clips per GPU, STFT+forward+decode" % (c["rnn_hidden_size"], B, c["seconds"]),
                       "clips_per_gpu": B, "clip_seconds": c["seconds"], "parallelism": "utterance-dp%d" % world,
                       "batches_in_flight": P},
            "roofline": roof,
            # every sampled kernel kind: mean dispatch time, ALGORITHMIC rates (SURVEY 8(d) FLOPs and bytes) and, where a
            # counter pass exists, the HBM/fabric bytes per launch it measured (FETCH_SIZE x2 + WRITE_SIZE) and that rate;
            # conv1/conv2 are the "conv front end" the north star asks GB/s for (HBM spec 8000 GB/s)
            "kernels": {k: {"avg_us": round(v["avg_us"], 2), "launches_per_step": v["launches"] // args.steps,
                            "tflops": round(v["flops_per_launch"] / (v["avg_us"] * 1e-6) / 1e12, 2),
                            "gbps": round(v["bytes_per_launch"] / (v["avg_us"] * 1e-6) / 1e9, 1),
                            "pmc_bytes": (PMC_TRAFFIC.get(k) or {}).get("bytes_per_launch"),
                            "pmc_gbps": (round(PMC_TRAFFIC[k]["bytes_per_launch"] / (v["avg_us"] * 1e-6) / 1e9, 1)
                                         if (PMC_TRAFFIC.get(k) or {}).get("bytes_per_launch") else None),
                            "mfma_busy": (PMC_TRAFFIC.get(k) or {}).get("mfma_busy")}
                        for k, v in sorted(stats.items()) if v["samples"] and v["avg_us"] > 0} if stats else None,
            "sample_transcript_len": len(out[0]) if out else None,
            "recomputed_batches": recomputed,
        }
        if world == 1 and not args.no_cpu_baseline:
            base, parity = cpu_baseline_and_parity(cfg, sd, B, n_samples, labels, last, out)
            result["cpu_baseline"] = base
            result.update(parity)
        else:
            result["cpu_baseline"] = None
            result["parity_checked"] = False
    for mdl in models:
        mdl.close()
    if rank == 0 and world == 1 and not args.no_public_surface:
        result["public_surface"] = public_surface(cfg, sd, B, n_samples, args.steps, args.warmup, out)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return result


def cpu_baseline_and_parity(cfg, sd, B, n_samples, labels, last, gpu_strings):
    """oracle/torch_port.py (kind "port": the reference's CPU operators, all physical cores) on the benchmarked batch,
    timed -- and the GPU's probabilities / transcripts of that same batch compared with it."""
    import torch
    from danspeech_amd import synthetic as syn
    from oracle import torch_port as tp, decoder as od
    cores, cpu_model = physical_cores()
    clips = [syn.make_clip(i, n_samples) for i in range(B)]
    # torch's CPU recurrent path is many small GEMMs per step: more threads than it can use make it SLOWER (128 threads: 17
    # audio-s/s, 8 vCPUs: 50).  Calibrate on a short sample of the same batch shape and time the full batch at the best count.
    probe = [c[:32000] for c in clips]
    xq, fq = tp.spectrogram_batch(probe)
    best, threads = None, cores
    for cand in sorted({c for c in (8, 16, 32, 64, cores) if c <= cores}):
        torch.set_num_threads(cand)
        tp.forward(sd, cfg, xq[:2], fq[:2])                        # thread pool + oneDNN primitive warm-up
        tq = time.perf_counter()
        tp.forward(sd, cfg, xq, fq)
        tq = time.perf_counter() - tq
        if best is None or tq < best:
            best, threads = tq, cand
    torch.set_num_threads(threads)
    tp.forward(sd, cfg, xq[:2], fq[:2])
    t0 = time.perf_counter()
    x, fr = tp.spectrogram_batch(clips)
    probs, out_lens = tp.forward(sd, cfg, x, fr)
    strings, _ = od.greedy_decode(probs, out_lens, labels, 0)
    dt = time.perf_counter() - t0
    base = {"value": round(B * n_samples / 16000.0 / dt, 2), "unit": "audio-s/s", "cores": threads, "kind": "port", "cpu": cpu_model,
            "host_physical_cores": cores,
            "sample": "one batch of %d x %.0f s clips through oracle/torch_port.py (F.conv2d / aten::gru / F.linear on %d threads, the "
                      "fastest of 8/16/32/64/%d on a 2 s probe of the same batch) + numpy STFT + greedy decode, %.1f s wall"
                      % (B, n_samples / 16000.0, threads, cores, dt)}
    pg = last["probs"].cpu().numpy()
    err = max(float(np.abs(pg[b, :out_lens[b]] - probs[b, :out_lens[b]]).max()) for b in range(B))
    same = sum(int(g == s[0]) for g, s in zip(gpu_strings, strings))
    parity = {"parity_checked": True, "max_err": err, "transcripts_identical": "%d/%d" % (same, B),
              "transcript_len_min_max": [min(len(s[0]) for s in strings), max(len(s[0]) for s in strings)]}
    if err >= 1e-4 or same != B or not np.array_equal(out_lens, last["out_lens"]):
        parity["parity_checked"] = "FAILED"
    return base, parity


def public_surface(cfg, sd, B, n_samples, steps, warmup, abi_strings):
    """The same workload through the drop-in surface: Recognizer(model=...).recognize_batches(float64 host arrays),
    i.e. host staging, PCIe upload, features, forward, decode, strings -- pipelined one batch ahead."""
    import torch
    from danspeech_amd import Recognizer, synthetic as syn
    from danspeech_amd.deepspeech.model import DeepSpeech
    import contextlib
    import io
    model = DeepSpeech("cfgA", rnn_type=cfg["rnn_type"], rnn_hidden_size=cfg["rnn_hidden_size"], rnn_layers=cfg["rnn_layers"],
                       conv_layers=cfg["conv_layers"]).load_state_dict(sd)
    with contextlib.redirect_stdout(io.StringIO()):
        rec = Recognizer(model=model)
    clips = [syn.make_clip(i, n_samples) for i in range(B)]
    for res in rec.recognize_batches([clips] * max(warmup, 1)):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for res in rec.recognize_batches([clips] * steps):
        pass
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(max(steps // 2, 1)):
        one = rec.recognize_batch(clips)
    dt1 = (time.perf_counter() - t1) / max(steps // 2, 1)
    return {"value": round(B * n_samples / 16000.0 * steps / dt, 2), "unit": "audio-s/s", "ms_per_step": round(dt / steps * 1e3, 3),
            "entry": "Recognizer.recognize_batches: float64 host arrays -> strings, staging + PCIe included, one batch of lookahead",
            "unpipelined_ms_per_step": round(dt1 * 1e3, 3),
            "same_strings_as_abi_path": bool(abi_strings is not None and res == abi_strings and one == abi_strings)}


if __name__ == "__main__":
    main()
