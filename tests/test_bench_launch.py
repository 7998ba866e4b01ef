"""bench.py --gpus N must work as the driver invokes it (no launcher): the parent starts the rank processes itself.
Checked here on the CPU with --dry-run (gloo, a stand-in engine): one JSON line, n_gpus = 2, every clip's transcript
gathered on rank 0; and a rank that dies takes the job down with a non-zero exit instead of leaving the others hanging."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=300, env=e)


def test_bench_starts_its_own_ranks_dry_run():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["transcripts_gathered"] == 2 * line["config"]["clips_per_gpu"]
    assert "dry-run" in line["data"] and line["roofline"] is None
    assert line["config"]["parallelism"] == "utterance-dp2"


def test_bench_eight_ranks_dry_run():
    """What the driver's scaling run invokes, `python bench.py --gpus 8 ...`: eight fresh rank processes, one line, every rank's
    transcripts gathered on rank 0 in every step."""
    r = _run(["--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["config"]["parallelism"] == "utterance-dp8"
    assert line["transcripts_gathered"] == 8 * line["config"]["clips_per_gpu"]


def test_bench_single_rank_dry_run_has_the_same_line_shape():
    r = _run(["--dry-run", "--steps", "2", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["transcripts_gathered"] == line["config"]["clips_per_gpu"]
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line


def test_a_dying_rank_fails_the_whole_launch():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"], env={"DSMI_BENCH_TEST_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_power_sampler_without_a_gpu_reports_nothing_and_does_not_raise():
    """bench.py's `energy` field on a host without the device's hwmon file and without rocm-smi: no reading, no exception (the line
    then carries nulls)."""
    import importlib.util
    import os
    import time
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ps = bench.PowerSampler(0, period=0.01)
    ps._files = [f for f in ps._files if os.path.exists(f)]
    with ps:
        time.sleep(0.05)
    w = ps.mean_watts()
    assert w is None or w > 0
    assert ps.count() >= 0
