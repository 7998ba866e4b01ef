#!/usr/bin/env python3
"""Headline benchmark: audio-seconds transcribed per wall-second (RTFx) of ``recognize()`` work -- spectrogram
features -> DeepSpeech forward -> greedy CTC decode -> strings -- on synthetic 10 s clips, batch 32 per GPU
(BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One process per GPU.  ``--gpus N`` with N > 1 starts the N rank processes itself (fresh children with RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set; the parent never touches a GPU), or runs as one rank
when those variables are already set (``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``).

A "step" is one pass of the hot path over one batch of 32 clips per GPU (weak scaling: per-GPU work is fixed).  The
timed path is the drop-in surface itself: ``Recognizer.recognize_batches`` (reference danspeech/Recognizer.py:82-95,
batched) is handed the rank's clips where they lie in HBM (``DeviceClips``: float64 PCM, the sample type load_audio hands to
recognize(), reference danspeech/audio/resources.py:640) and returns strings on the host; two batches are in flight.
Inputs are resident before the timed region: at N > 1 rank 0 synthesises all clips and scatters the shards over RCCL as
int16 (widened on the device before the timed region); the per-step gather of the transcripts to rank 0 (fixed-size
payload) is inside the timed region.  Weights are seeded random tensors of the DanSpeech shapes (no network for
checkpoints): data = "synthetic"; they are scaled so that a 10 s clip decodes to 100+ tokens (synthetic.TALKATIVE), and
the timed batch is CHECKED against the CPU oracle -- a failed check makes the process exit non-zero.

Rank 0 prints ONE JSON line (see the driver contract), including
  roofline       -- the kernel with the largest total time in the timed region: algorithmic FLOPs per
                    launch / its mean dispatch duration (per-dispatch begin/end timestamps sampled live
                    through hipExtLaunchKernelGGL events) vs the MFMA peak of the instruction it runs on
  cpu_baseline   -- oracle/torch_port.py (the reference's own CPU operators: oneDNN conv, aten::gru, ...;
                    kind "port") on this host's physical cores, same batch
  parity         -- max |probs - oracle| and transcript equality of the GPU's batch vs that oracle run
  host_arrays    -- the same call with float64 HOST arrays in (staging + PCIe upload included), N = 1 only; never `value`
  abi_path       -- the same work as bare C-ABI calls (dsmi_features / dsmi_forward / dsmi_greedy), N = 1 only
  steady_state   -- the timed entry again over --steady-steps (>= 96) steps after the timed region, with `energy` (mean board
                    power and joules per batch over those steps), N = 1 only; never `value`
  rccl           -- N > 1: RCCL's version and the world size the process group reports (a rank count other than --gpus exits 4)

``--dry-run`` (CPU, no GPU work, backend gloo, a stand-in engine) exercises the launch, scatter, per-step gather and
the JSON line; its numbers mean nothing and the line says so.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # one hardware queue per stream of the pipeline (the engine asks for the same: _native.want_hw_queues)

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
# separate passes on this workload (profiles/*pmc*.md, tools/pmc_summary.py); None where no pass has been run.
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


def launch_ranks(n, argv):
    """``--gpus n`` without a launcher: start the n rank processes (fresh interpreters, never a re-exec of a process that has
    touched a GPU), wait for them, end the others if one fails, and exit with the worst return code.  Rank 0's stdout is
    read here: its JSON line becomes this command's stdout, anything else a library prints there (gloo's and RCCL's
    connection notes) goes to stderr."""
    import socket
    import subprocess
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))

    def forward(pipe):
        for line in pipe:
            (sys.stdout if line.startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()

    reader = threading.Thread(target=forward, args=(procs[0].stdout,), daemon=True)
    reader.start()
    worst = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                worst = worst or (rc if rc > 0 else 128 - rc)
                for q in live:                                 # the others would wait in a collective for ever
                    q.terminate()
    reader.join(timeout=10)
    return worst


class _DryEngine(object):
    """--dry-run: stands in for the recogniser.  A clip's "transcript" is a function of its samples."""

    def recognize_batches(self, batches):
        for clips in batches:
            rows = clips.pcm.view(len(clips), -1).to("cpu") if hasattr(clips, "pcm") else clips
            yield ["dry %d" % int(row.sum()) for row in rows]


class PowerSampler(object):
    """Board power of one GPU, sampled by a thread while a block runs.  Source: the hwmon `power1_input` / `power1_average` file
    (microwatts, label PPT) of the device's PCI function; a box shows the files of every GPU of its node, so the device is
    looked up by its PCI address (torch's device properties), and where that fails the file that reads highest over the block
    is taken -- the one GPU this process loads.  `rocm-smi --showpower` where no file is readable (slower: fewer samples)."""

    def __init__(self, index=0, period=0.05):
        self.index, self.period, self.vals, self.source = index, period, {}, None
        self._stop = None
        self._files = self._find_hwmon(index)

    @staticmethod
    def _find_hwmon(index):
        import glob
        def files_under(dev):
            for name in ("power1_average", "power1_input"):
                hits = glob.glob(os.path.join(dev, "hwmon", "hwmon*", name))
                if hits:
                    return hits[:1]
            return []
        try:                                                     # the device's own PCI function
            import torch
            pr = torch.cuda.get_device_properties(index)
            addr = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            hit = files_under("/sys/bus/pci/devices/" + addr)
            if hit:
                return hit
        except Exception:
            pass
        out = []
        for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            out += files_under(dev)
        return out

    def _read_all(self):
        got = {}
        for f in list(self._files):
            try:
                with open(f) as fh:
                    got[f] = int(fh.read().strip()) / 1e6
            except (OSError, ValueError):
                self._files.remove(f)
        if got:
            return got
        import re
        import subprocess
        try:
            out = subprocess.run(["rocm-smi", "-d", str(self.index), "--showpower"], capture_output=True, text=True, timeout=5).stdout
            m = re.search(r"Power[^:]*:\s*([0-9.]+)", out)
            return {"rocm-smi --showpower": float(m.group(1))} if m else {}
        except Exception:
            return {}

    def __enter__(self):
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                for k, v in self._read_all().items():
                    self.vals.setdefault(k, []).append(v)
                self._stop.wait(self.period)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._thread.join(timeout=10)
        if self.vals:                                            # several candidates: the one this process loaded reads highest
            best = max(self.vals, key=lambda k: sum(self.vals[k]) / len(self.vals[k]))
            self.source = ("sysfs " + best if best.startswith("/") else best) + (" (highest of %d candidates)" % len(self.vals) if len(self.vals) > 1 else "")
            self._best = self.vals[best]
        else:
            self._best = []

    def mean_watts(self):
        return sum(self._best) / len(self._best) if getattr(self, "_best", None) else None

    def count(self):
        return len(getattr(self, "_best", []) or [])


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


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)     # the timed region holds the fill and the drain of a pipeline of four forwards
                                                         # of two batches each: 40 steps read 8 % low, 96 steps 3 %
    ap.add_argument("--warmup", type=int, default=16)    # at least 16 untimed steps are run whatever is asked for (every lane's
                                                         # workspaces and pinned staging slots are allocated on their first use)
    ap.add_argument("--config", default="cfgA-greedy", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default: the config's 32)")
    ap.add_argument("--hidden", type=int, default=None, help="experiments: another hidden size (the JSON line then names it)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle run (and with it the parity check)")
    ap.add_argument("--no-side-paths", action="store_true", help="skip device_resident, abi_path and f32_strict")
    ap.add_argument("--timed-input", default="host", choices=["host", "device"],
                    help="profiling only: time the device-resident entry instead of the metric's host-array entry (the line says so)")
    ap.add_argument("--strict-f32-child", action="store_true", help=argparse.SUPPRESS)     # the f32_strict side run (a fresh process)
    ap.add_argument("--abi-child", action="store_true", help=argparse.SUPPRESS)            # the abi_path side run (a fresh process)
    ap.add_argument("--other-config-child", type=int, default=0, help=argparse.SUPPRESS)   # one of other_configs (a fresh process)
    ap.add_argument("--warmup-calls", type=int, default=1, help=argparse.SUPPRESS)          # experiments: the warm-up as this many calls
    ap.add_argument("--no-other-configs", action="store_true", help="skip the side runs of BASELINE.json configs 3, 4 and 5's per-GPU share")
    ap.add_argument("--no-kernel-sampling", action="store_true")
    ap.add_argument("--lanes", type=int, default=None, help="experiments: forwards in flight (default: the engine's own choice; the line says what ran)")
    ap.add_argument("--steady-steps", type=int, default=384, help="steps of the steady_state side run (at least 96)")
    ap.add_argument("--dry-run", action="store_true", help="CPU only: launch, scatter, gather and the JSON line with a stand-in engine")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, argv))
    if args.other_config_child:
        print(json.dumps(other_config(args.other_config_child)), flush=True)
        return 0

    import torch
    import torch.distributed as dist
    from danspeech_amd import synthetic as syn
    from danspeech_amd import parallel
    from danspeech_amd.audio.parsers import DeviceClips

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dry = args.dry_run
    if dry and os.environ.get("DSMI_BENCH_TEST_FAIL_RANK") == str(rank):      # tests/test_bench_launch.py: a rank that dies at start-up
        raise SystemExit(3)
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    rccl = None
    # (DSMI_BENCH_FORCE_GROUP=1: the process group, the barrier, the reduction and the gather stream of the N > 1 path with ONE rank --
    # the boxes of this pool have one GPU; tests/test_gpu_rccl_one_rank.py)
    multi = world > 1 or bool(os.environ.get("DSMI_BENCH_FORCE_GROUP"))
    if multi and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    line_fd = None
    if multi and not dry:
        # RCCL writes notes of its own to the C-level stdout ("Librccl path : ...", buffered until the process ends: BEHIND the JSON
        # line).  The line is this command's stdout; everything else that is written to descriptor 1 from here on goes to stderr.
        sys.stdout.flush()
        line_fd = os.dup(1)
        os.dup2(2, 1)
    if multi:
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:          # a launcher that started fewer ranks than asked for must not read as N GPUs
            sys.stderr.write("bench.py: --gpus %d but the process group has %d ranks\n" % (args.gpus, dist.get_world_size()))
            raise SystemExit(4)
        if not dry:
            rccl = {"version": ".".join(str(x) for x in torch.cuda.nccl.version()), "world_size_seen": dist.get_world_size(),
                    "backend": dist.get_backend()}

    c = dict(CONFIGS[args.config])
    if args.hidden:
        c["rnn_hidden_size"] = args.hidden
    B = args.batch or c["batch"]
    n_samples = int(c["seconds"] * 16000) if not dry else 1600
    cfg = {k: c[k] for k in ("conv_layers", "rnn_type", "rnn_hidden_size", "rnn_layers", "bidirectional", "context")}
    labels = syn.DANSPEECH_LABELS
    handles = []
    if args.abi_child:              # only the bare C-ABI loop, in a process that has made no other stream
        import torch
        sd = syn.make_state_dict(cfg["conv_layers"], cfg["rnn_type"], cfg["rnn_hidden_size"], cfg["rnn_layers"],
                                 bidirectional=cfg["bidirectional"], seed=0, **syn.TALKATIVE)
        r = abi_path(cfg, sd, B, n_samples, args.steps, args.warmup, labels, None, torch.device("cuda:0"))
        print(json.dumps(r), flush=True)
        return 0
    if dry:
        rec, eng, sd = _DryEngine(), None, None
    else:
        import contextlib
        import io
        from danspeech_amd import Recognizer
        from danspeech_amd.deepspeech.model import DeepSpeech
        sd = syn.make_state_dict(cfg["conv_layers"], cfg["rnn_type"], cfg["rnn_hidden_size"], cfg["rnn_layers"],
                                 bidirectional=cfg["bidirectional"], seed=0, **syn.TALKATIVE)
        model = DeepSpeech("cfgA", rnn_type=cfg["rnn_type"], rnn_hidden_size=cfg["rnn_hidden_size"], rnn_layers=cfg["rnn_layers"],
                           conv_layers=cfg["conv_layers"]).load_state_dict(sd)
        with contextlib.redirect_stdout(io.StringIO()):
            rec = Recognizer(model=model)                # greedy decoding: no language model
        eng = rec.danspeech_recognizer
        eng.keep_last_output = True
        if args.lanes:
            eng.pipeline_lanes = args.lanes

    # ---- inputs: rank 0 synthesises, shards go out over RCCL as int16 (utterance-level data parallelism)
    if rank == 0:
        all_clips = np.stack([syn.make_clip(i, n_samples) for i in range(B * world)])   # integer-valued float64 [B*world, N]
    else:
        all_clips = None
    pcm = parallel.scatter_clips(all_clips, B, n_samples, rank, world, dev, dtype=np.int16).to(torch.float64)
    clips = DeviceClips(pcm.view(-1), np.full(B, n_samples, dtype=np.int64))
    # what the metric's entry takes: float64 HOST arrays, as load_audio returns them (reference Recognizer.py:82-95, resources.py:640);
    # at N > 1 every rank times the same entry on the shard it received
    host_clips = [row for row in pcm.view(B, -1).cpu().numpy()] if not dry else clips
    if args.timed_input == "device":
        host_clips = clips
    cap = max((n_samples // 160 + 1 + 1) // 2, 16)   # a transcript is never longer than the OUTPUT frame count (time stride 2)
    positions = np.arange(rank * B, (rank + 1) * B)

    # the gather gets a stream of its own: RCCL orders a collective behind everything already queued on the stream it is issued
    # from, and the pipeline's first stream always holds a forward that was enqueued ahead (8 ms of somebody else's work)
    import contextlib
    gather_stream = torch.cuda.Stream(device=local) if (multi and not dry) else None

    def run(steps):
        """`steps` batches through recognize_batches; the transcripts of every step gathered to rank 0."""
        out = None
        # (a LIST of `steps` batches, as a caller with its clips in hand passes them: the pipeline then knows where the call ends and
        # deals its last round of forwards evenly over the lanes -- 20 steps are eight forwards of 64 clips and four of 32)
        for res in rec.recognize_batches([host_clips] * steps):
            if multi:
                with (torch.cuda.stream(gather_stream) if gather_stream is not None else contextlib.nullcontext()):
                    out = parallel.gather_texts(res, positions, B * world, cap, rank, world, dev)
            else:
                out = res
        return out

    def sync():
        if multi:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    # ONE warm-up call of at least 16 steps (every lane's workspaces and pinned staging slots are made on their first use).  Round 5
    # needed two: a process's second call paid a blocking first upload on each replica lane -- the first copy a process hands a DMA
    # engine holds the caller until it is done; the engine now meets its engines when its lanes are set up
    # (DanSpeechRecognizer._warm_copy_engines, profiles/r06_second_call_stall.txt)
    # (+ steps % 8: the call's last round of forwards then has the shape of the timed call's -- 20 timed steps end in four 32-clip
    # forwards, and a forward of a size the process has not seen costs its first call tensors of new sizes: 27 ms in a 20-step call)
    warmup_each = max(args.warmup, 16 if not dry else 0) + (args.steps % 8 if not dry else 0)
    out = None
    for _ in range(max(args.warmup_calls, 1)):
        out = run(warmup_each) or out
    warmup_done = warmup_each * max(args.warmup_calls, 1)
    if eng is not None:
        handles = [eng.model._native] + [r[0]._native for r in eng._replicas]
    # Per-dispatch timestamps on ONE of the forwards' lanes: a stamped launch is a hipExtLaunchKernelGGL with two events and costs the
    # enqueueing thread ~0.2 ms; with every lane stamped (60 recurrent + a dozen dense launches in a 20-step call) that was 0.65 ms per
    # step of the region being timed (6.35 against 5.71, r6z).  The lanes are dealt the forwards in turn and run the same kernels: one
    # lane's launches are a quarter of every kind's, and the launch counts below are scaled by the number of lanes.
    sampled = handles[:1] if not args.no_kernel_sampling else []
    for h in sampled:
        h.set_profiling(2)
        h.reset_kernel_stats()
    sync()
    t0 = time.perf_counter()
    out = run(args.steps) or out
    sync()
    dt = time.perf_counter() - t0
    stats = {}
    if sampled:
        for h in sampled:                        # merge the sampled handles' figures
            for k, v in h.kernel_stats().items():
                a = stats.setdefault(k, dict(launches=0, samples=0, _us=0.0, _fl=0.0, _by=0.0))
                a["launches"] += v["launches"]; a["samples"] += v["samples"]
                a["_us"] += v["avg_us"] * v["samples"]; a["_fl"] += v["flops_per_launch"] * v["launches"]
                a["_by"] += v["bytes_per_launch"] * v["launches"]
            h.set_profiling(0)
        for a in stats.values():
            a["avg_us"] = a["_us"] / max(a["samples"], 1)
            a["flops_per_launch"] = a["_fl"] / max(a["launches"], 1)
            a["bytes_per_launch"] = a["_by"] / max(a["launches"], 1)
            a["launches"] *= len(handles) / float(len(sampled))
    # ---- side measurement, after the timed region: the same entry over enough steps that the pipeline's fill and drain (four forwards
    # of two batches) are a few per cent of it, with the board power sampled beside it.  Never `value`.
    steady = None
    if world == 1 and not dry and not args.no_side_paths and not args.strict_f32_child:
        ss_steps = max(args.steady_steps, 96)
        with PowerSampler(local) as pw:
            sync()
            t1 = time.perf_counter()
            run(ss_steps)
            sync()
            dts = time.perf_counter() - t1
        steady = {"value": round(B * (n_samples / 16000.0) * ss_steps / dts, 2), "unit": "audio-s/s", "ms_per_step": round(dts / ss_steps * 1e3, 3),
                  "steps": ss_steps, "entry": "the timed entry again, run after the timed region"}
        watts = pw.mean_watts()
        steady["energy"] = {"watts_mean": None if watts is None else round(watts, 1),
                            "joules_per_batch": None if watts is None else round(watts * dts / ss_steps, 3),
                            "samples": pw.count(), "source": pw.source, "note": "board power while the steady-state steps ran"}
    recomputed = sum(h.recompute_count() for h in handles)
    P = (1 + len(eng._replicas)) if eng is not None else 2           # forwards in flight, each of up to pipeline_merge_clips clips
    merge_clips = max(eng.pipeline_merge_clips, B) if eng is not None else B

    if multi:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    result = None
    failed = False
    if rank == 0:
        audio_s = world * B * (n_samples / 16000.0) * args.steps
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
                        # `achieved` / `frac` are per launch (the contract's definition).  Launches of this kernel overlap (one per
                        # forward in flight, each on its own CUs): their summed durations / the timed region's wall time is the
                        # MEASURED time-averaged number of them running, and frac x that the share of the chip's peak this kernel
                        # sustains over the whole region
                        mean_concurrent_launches=round(s["avg_us"] * 1e-6 * s["launches"] / dt, 3),
                        frac_chip_time_averaged=round(ach / peak * s["avg_us"] * 1e-6 * s["launches"] / dt, 4),
                        traffic_source=PMC_TRAFFIC.get("_source", "profiles/pmc_traffic.json (builder's counter pass of an earlier tree, "
                                                                   "not measured in this run)"),
                        avg_launch_us=round(s["avg_us"], 3), launches_per_step=round(s["launches"] / args.steps, 3),
                        sampled_launches=int(s["samples"]), sampling="every launch of this kernel on one of the %d lanes (the lanes are dealt the "
                        "forwards in turn); launch counts scaled by the lanes" % max(len(handles), 1),
                        flops_per_launch=s["flops_per_launch"],
                        kernel_time_share={k: round(v / sum(tot.values()), 4) for k, v in sorted(tot.items())})
        result = {
            "metric": "audio-seconds/sec (RTFx) recognize() on 10 s clips, batch=32",
            "value": round(value, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_done": warmup_done,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (two-term fp16 split operands, 3 MFMA products, fp32 accumulate)",
            "data": "synthetic" if not dry else "dry-run (CPU stand-in engine: NOT a measurement)",
            "config": {"workload": "BASELINE.json configs[1]: 2conv + 5xBiGRU%d (DanSpeechPrimary per BASELINE), greedy CTC, "
                                   "batch=%d x %.0f s 16 kHz clips per GPU, STFT+forward+decode" % (c["rnn_hidden_size"], B, c["seconds"]),
                       "entry": ("Recognizer.recognize_batches: float64 HOST arrays -> strings on the host (pinned staging and the PCIe "
                                 "upload inside the timed region); the caller hands over batches of %d clips" % B) if args.timed_input == "host"
                                else "PROFILING RUN, not the metric's entry: Recognizer.recognize_batches(DeviceClips), float64 PCM resident in HBM",
                       "clips_per_gpu": B, "clip_seconds": n_samples / 16000.0, "parallelism": "utterance-dp%d" % world,
                       "forwards_in_flight": P, "clips_per_forward": merge_clips,
                       "batches_in_flight": P * max(merge_clips // B, 1),
                       "last_round": "the call's last round of forwards is dealt evenly over the lanes (a sized source): the timed call's "
                                     "launches are of %d clips, its last %s of fewer -- roofline.flops_per_launch and avg_launch_us "
                                     "are means over all of them" % (merge_clips, "round's") if eng is not None and eng.pipeline_balance_tail else None},
            "roofline": roof,
            # every sampled kernel kind: mean dispatch time, ALGORITHMIC rates (SURVEY 8(d) FLOPs and bytes) and, where a
            # counter pass exists, the HBM/fabric bytes per launch it measured (FETCH_SIZE x2 + WRITE_SIZE) and that rate;
            # conv1/conv2 are the "conv front end" the north star asks GB/s for (HBM spec 8000 GB/s)
            "kernels": {k: {"avg_us": round(v["avg_us"], 2), "launches_per_step": round(v["launches"] / args.steps, 3),
                            "tflops": round(v["flops_per_launch"] / (v["avg_us"] * 1e-6) / 1e12, 2),
                            "gbps": round(v["bytes_per_launch"] / (v["avg_us"] * 1e-6) / 1e9, 1),
                            "pmc_bytes": (PMC_TRAFFIC.get(k) or {}).get("bytes_per_launch"),
                            "pmc_gbps": (round(PMC_TRAFFIC[k]["bytes_per_launch"] / (v["avg_us"] * 1e-6) / 1e9, 1)
                                         if (PMC_TRAFFIC.get(k) or {}).get("bytes_per_launch") else None),
                            "mfma_busy": (PMC_TRAFFIC.get(k) or {}).get("mfma_busy")}
                        for k, v in sorted(stats.items()) if v["samples"] and v["avg_us"] > 0} if stats else None,
            "sample_transcript_len": len(out[0]) if out else None,
            "transcripts_gathered": len(out) if out else 0,
            "recomputed_batches": recomputed,
            "steady_state": steady,
            "energy": steady["energy"] if steady else None,
        }
        if rccl:
            result["rccl"] = rccl
        if world == 1 and not args.no_cpu_baseline and not dry:
            probs, sizes = eng.last_output          # the last forward: consecutive batches merged, this batch's clips first
            base, parity = cpu_baseline_and_parity(cfg, sd, B, n_samples, labels, {"probs": probs[:B], "out_lens": np.asarray(sizes)[:B]}, out)
            result["cpu_baseline"] = base
            result.update(parity)
            failed = parity["parity_checked"] == "FAILED"
        else:
            result["cpu_baseline"] = None
            result["parity_checked"] = False
        if world == 1 and not args.no_side_paths and not dry:
            result["device_resident"] = device_resident(rec, clips, host_clips, B, n_samples, args.steps, out)
            failed = failed or not result["device_resident"]["same_strings_as_timed_path"]
    if rank == 0 and world == 1 and not args.no_side_paths and not dry:
        del rec, eng
        result["abi_path"] = abi_path_child(args, out)
        failed = failed or not result["abi_path"]["same_strings_as_timed_path"]
        if not args.strict_f32_child:
            result["f32_strict"] = f32_strict_child(args)
        if not args.strict_f32_child and not args.no_other_configs:
            result["other_configs"] = other_configs_children()
    if rank == 0:
        if line_fd is None:
            print(json.dumps(result), flush=True)
        else:
            sys.stdout.flush()
            os.write(line_fd, (json.dumps(result) + "\n").encode())
    if multi:
        dist.destroy_process_group()
    if failed:
        sys.exit(2)           # a throughput line from a computation that failed its own check must not read as a result
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
    # audio-s/s, 8 vCPUs: 50).  Fixed policy: 16 and 32 threads (what the host has of them), warm-up outside the timed region, the
    # MEDIAN of three runs of the whole batch at each count; `value` is the better of the two medians, both are reported.
    runs = {}
    probs = out_lens = strings = None
    for threads in sorted({min(t, cores) for t in (16, 32)}):
        torch.set_num_threads(threads)
        xq, fq = tp.spectrogram_batch([c[:32000] for c in clips[:4]])
        tp.forward(sd, cfg, xq, fq)                                 # thread pool + oneDNN primitive warm-up
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            x, fr = tp.spectrogram_batch(clips)
            probs, out_lens = tp.forward(sd, cfg, x, fr)
            strings, _ = od.greedy_decode(probs, out_lens, labels, 0)
            times.append(time.perf_counter() - t0)
        runs[threads] = sorted(times)[1]
    threads = min(runs, key=runs.get)
    dt = runs[threads]
    base = {"value": round(B * n_samples / 16000.0 / dt, 2), "unit": "audio-s/s", "cores": threads, "kind": "port", "cpu": cpu_model,
            "host_physical_cores": cores,
            "by_threads": {str(k): round(B * n_samples / 16000.0 / v, 2) for k, v in sorted(runs.items())},
            "sample": "one batch of %d x %.0f s clips through oracle/torch_port.py (F.conv2d / aten::gru / F.linear) + numpy STFT + greedy "
                      "decode; median of 3 runs at each of %s threads after a warm-up, the better median is `value` (%.1f s per run)"
                      % (B, n_samples / 16000.0, "/".join(str(k) for k in sorted(runs)), dt)}
    pg = last["probs"].cpu().numpy()
    err = max(float(np.abs(pg[b, :out_lens[b]] - probs[b, :out_lens[b]]).max()) for b in range(B))
    same = sum(int(g == s[0]) for g, s in zip(gpu_strings, strings))
    parity = {"parity_checked": True, "max_err": err, "transcripts_identical": "%d/%d" % (same, B),
              "transcript_len_min_max": [min(len(s[0]) for s in strings), max(len(s[0]) for s in strings)]}
    if err >= 1e-4 or same != B or not np.array_equal(out_lens, last["out_lens"]):
        parity["parity_checked"] = "FAILED"
    return base, parity


def device_resident(rec, clips, host_clips, B, n_samples, steps, timed_strings):
    """The same call with the clips already in HBM (DeviceClips, what an RCCL scatter delivers): no staging, no PCIe.  And one
    call at a time (`recognize_batch`) on the host arrays.  Reported beside `value`, never as it."""
    import torch
    for res in rec.recognize_batches([clips] * 8):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for res in rec.recognize_batches([clips] * steps):
        pass
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(max(steps // 4, 1)):
        one = rec.recognize_batch(host_clips)
    dt1 = (time.perf_counter() - t1) / max(steps // 4, 1)
    return {"value": round(B * n_samples / 16000.0 * steps / dt, 2), "unit": "audio-s/s", "ms_per_step": round(dt / steps * 1e3, 3),
            "entry": "Recognizer.recognize_batches(DeviceClips): float64 PCM resident in HBM -> strings on the host",
            "unpipelined_host_arrays_ms_per_step": round(dt1 * 1e3, 3),
            "same_strings_as_timed_path": bool(timed_strings is not None and res == timed_strings and one == timed_strings)}


# BASELINE.json configs 3-5 as side figures of the line (never `value`): SURVEY 8(d) shapes, synthetic weights, a seeded synthetic ARPA
# language model built in a temporary directory (the DSL .klm files are unobtainable offline).  flops = SURVEY 8(d)'s algorithmic
# figure per clip (cfgA 10 s 51.85, cfgB 10 s 132.9, cfgA 30 s 155.4 GFLOP).
OTHER_CONFIGS = {
    3: dict(key="config3", what="BASELINE.json configs[2]: cfgA (5 x BiGRU 800) + 3-gram, CTC beam 64, batch 32 x 10 s",
            hidden=800, layers=5, lm_order=3, beam=64, alpha=1.3, beta=0.2, batch=32, seconds=10.0, gflop_per_clip=51.85, warm=32, steps=96),
    4: dict(key="config4", what="BASELINE.json configs[3]: 7 x BiGRU 1200 + 5-gram, CTC beam 128, batch 64 x 10 s",
            hidden=1200, layers=7, lm_order=5, beam=128, alpha=1.3, beta=0.2, batch=64, seconds=10.0, gflop_per_clip=132.9, warm=12, steps=48),
    5: dict(key="config5_share", what="BASELINE.json configs[4], ONE GPU's share of 1024 x 30 s over 8 GPUs: cfgA + 3-gram, CTC beam 64, "
            "batches of 128 x 30 s", hidden=800, layers=5, lm_order=3, beam=64, alpha=1.3, beta=0.2, batch=128, seconds=30.0,
            gflop_per_clip=155.4, warm=6, steps=16),
}


def other_config(no):
    """One of OTHER_CONFIGS through the metric's entry (Recognizer.recognize_batches, float64 host arrays -> every clip's best beam as a
    string), as a stream of batches; in a process of its own (see abi_path_child on streams and hardware queues)."""
    import contextlib
    import io
    import tempfile
    import torch
    from danspeech_amd import Recognizer, synthetic as syn
    from danspeech_amd.deepspeech.model import DeepSpeech
    c = OTHER_CONFIGS[no]
    sd = syn.make_state_dict(2, "gru", c["hidden"], c["layers"], seed=0, **syn.TALKATIVE)
    model = DeepSpeech("cfg", rnn_hidden_size=c["hidden"], rnn_layers=c["layers"], conv_layers=2).load_state_dict(sd)
    with tempfile.TemporaryDirectory() as td, contextlib.redirect_stdout(io.StringIO()):
        rec = Recognizer(model=model)
        lm = os.path.join(td, "syn%d.arpa" % c["lm_order"])
        syn.make_arpa(lm, order=c["lm_order"], n_words=5000, seed=11, ngrams_per_order=20000)
        rec.update_decoder(lm=lm, alpha=c["alpha"], beta=c["beta"], beam_width=c["beam"])
        B, n = c["batch"], int(c["seconds"] * 16000)
        clips = [syn.make_clip(i, n) for i in range(B)]
        one = rec.recognize_batch(clips)
        for res in rec.recognize_batches([clips] * c["warm"]):      # every lane's workspaces, staging slots and decoder slots exist
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        same = True
        for res in rec.recognize_batches([clips] * c["steps"]):
            same = same and res == one
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / c["steps"]
    eng = rec.danspeech_recognizer
    handles = [eng.model._native] + [r[0]._native for r in eng._replicas]
    tflop = c["gflop_per_clip"] * B / 1e3
    return {"workload": c["what"], "ms_per_batch": round(dt * 1e3, 3), "audio_s_per_s": round(B * c["seconds"] / dt, 1), "steps": c["steps"],
            "whole_step_tflops": round(tflop / dt, 1), "whole_step_frac": round(tflop / dt / PEAK_SPLIT_TFLOPS, 4),
            "forwards_in_flight": len(handles), "clips_per_forward": min(B, eng.pipeline_merge_clips),
            "recomputed_batches": sum(h.recompute_count() for h in handles),
            "every_batch_equals_the_single_call": bool(same), "sample_transcript_len": len(one[0]),
            "lm": "synthetic %d-gram ARPA, 5000 words (danspeech_amd.synthetic.make_arpa, seed 11)" % c["lm_order"]}


def other_configs_children():
    """other_config(3 | 4 | 5), each in a fresh process (this one keeps the persistent kernels' per-device lock: DSMI_PERSIST_SHARED)."""
    import subprocess
    res = {"note": "side figures, never `value`: the other BASELINE.json configs through Recognizer.recognize_batches (host arrays in, best "
                   "beam per clip out) as a stream of batches; whole_step_frac = SURVEY 8(d) algorithmic FLOPs per batch / time / 833 TFLOP/s"}
    env = dict(os.environ, DSMI_PERSIST_SHARED="1")
    for no, c in sorted(OTHER_CONFIGS.items()):
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--other-config-child", str(no)], env=env, capture_output=True,
                                 text=True, timeout=600)
            res[c["key"]] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        except Exception as e:      # (a side figure: reported as failed, the line still comes out)
            res[c["key"]] = {"ms_per_batch": None, "error": repr(e)[:300]}
    return res


def f32_strict_child(args):
    """The strict reading of the reference's arithmetic (fp32 operands on the fp32 MFMA, one launch per recurrent step:
    DSMI_DENSE_MODE=f32 DSMI_RNN_MODE=steps) through the same entry, as a fresh process; parity-checked like the main run."""
    import subprocess
    env = dict(os.environ, DSMI_DENSE_MODE="f32", DSMI_RNN_MODE="steps")
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(max(args.steps // 4, 4)), "--warmup", "2", "--config", args.config,
           "--no-kernel-sampling", "--strict-f32-child", "--no-side-paths"]
    if args.batch:
        cmd += ["--batch", str(args.batch)]
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        r = json.loads(line)
        return {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                "dtype": "f32 (fp32 operands, v_mfma_f32_32x32x2_f32, one launch per recurrent step)",
                "parity_checked": r.get("parity_checked"), "max_err": r.get("max_err"), "transcripts_identical": r.get("transcripts_identical")}
    except Exception as e:          # a side figure must not take the line down with it
        return {"value": None, "error": repr(e)[:300]}


def abi_path(cfg, sd, B, n_samples, steps, warmup, labels, timed_strings, dev):
    """The same work as bare C-ABI calls -- dsmi_features / dsmi_forward / dsmi_forward_status / dsmi_greedy_enqueue / _collect on four
    handle sets and four streams, TWO of the caller's batches per call (64 clips: what the Python engine merges) -- without the Python
    engine between them: what a host in another language gets."""
    import torch
    from danspeech_amd import _native, synthetic as syn
    P = 4
    local = dev.index or 0
    models = [_native.NativeModel(cfg, sd, device=local, n_labels=len(labels)) for _ in range(P)]
    for mdl in models:
        mdl.set_inflight(P)
    frontends = [_native.NativeFrontend(device=local) for _ in range(P)]
    decoders = [_native.NativeDecoder(labels, blank_index=0, device=local) for _ in range(P)]
    streams = [torch.cuda.Stream(device=local) for _ in range(P)]
    M = 2                                            # batches per call
    pcm = torch.from_numpy(np.stack([syn.make_clip(i % B, n_samples) for i in range(M * B)])).to(dev)
    n = np.full(M * B, n_samples, dtype=np.int64)
    inflight, step_no = [], [0]

    def finish(item):
        k, probs, out_lens = item
        models[k].status()
        dec = decoders[k].greedy_collect()
        return ["".join(labels[i] for i in d[0]) for d in dec][-B:]

    def step():
        k = step_no[0] % P
        step_no[0] += 1
        with torch.cuda.stream(streams[k]):
            feat, fr = frontends[k].features(pcm.view(-1), n)
            probs, out_lens = models[k].forward(feat, fr, check=False)
            decoders[k].greedy_enqueue(probs, out_lens)
        inflight.append((k, probs, out_lens))
        return finish(inflight.pop(0)) if len(inflight) >= P else None

    def drain():
        out = None
        while inflight:
            out = finish(inflight.pop(0))
        return out

    out = None
    steps = max(steps // M, 1)                       # a step of this loop carries M of the caller's batches
    for _ in range(max(warmup // M, 2 * P)):
        out = step() or out
    out = drain() or out
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step() or out
    out = drain() or out
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for mdl in models:
        mdl.close()
    return {"value": round(M * B * n_samples / 16000.0 * steps / dt, 2), "unit": "audio-s/s", "ms_per_step": round(dt / (steps * M) * 1e3, 3),
            "entry": "dsmi_features + dsmi_forward + dsmi_forward_status + dsmi_greedy_enqueue/_collect, float64 PCM resident in HBM, four calls "
                     "of two 32-clip batches each in flight; a process of its own (four streams, nothing else: a stream that shares a "
                     "hardware queue with another runs behind it)",
            "same_strings_as_timed_path": bool(timed_strings is not None and out == timed_strings), "strings": out}


def abi_path_child(args, timed_strings):
    """abi_path in a fresh process: the ROCm runtime deals streams onto GPU_MAX_HW_QUEUES hardware queues in turn, and behind the
    engine's streams of the timed run two of the four streams here landed on one queue as often as not (33-47 k audio-s/s from run
    to run, 42-55 k for repeated calls in one process)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--config", args.config,
           "--abi-child", "--no-side-paths", "--no-cpu-baseline", "--no-kernel-sampling"]
    if args.batch:
        cmd += ["--batch", str(args.batch)]
    if args.hidden:
        cmd += ["--hidden", str(args.hidden)]
    # (this process keeps the per-device lock of the persistent kernels until it exits and runs nothing while it waits: the child
    # is told not to ask for the lock -- without it it would take the one-launch-per-step path, 18 k audio-s/s)
    env = dict(os.environ, DSMI_PERSIST_SHARED="1")
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        r["same_strings_as_timed_path"] = bool(timed_strings is not None and r.pop("strings", None) == timed_strings)
        return r
    except Exception as e:          # (a side figure: reported as failed, the line still comes out)
        return {"value": None, "error": repr(e)[:300], "same_strings_as_timed_path": False}


if __name__ == "__main__":
    main()
