"""bench.py contract tests.  CPU: `--gpus 2` spawns two ranks by itself (no external launcher), rendezvous over gloo on
127.0.0.1, reports n_gpus == 2; a launcher / --gpus mismatch fails loudly.  GPU (1-GPU box): the full two-rank bench with
both ranks on device 0 over gloo, including the chunked solve + all_gather measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    if env:
        e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True,
                          cwd=ROOT, env=e, timeout=timeout)


def last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert lines, stdout
    return json.loads(lines[-1])


def test_gpus_flag_spawns_the_ranks_itself():
    r = run_bench("--gpus", "2", "--backend", "gloo", "--plumbing-only", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    out = last_json(r.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1


def test_world_size_mismatch_fails_loudly():
    r = run_bench("--gpus", "4", "--plumbing-only", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_two_rank_bench_on_one_gpu():
    r = run_bench("--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "10", "--warmup", "3", "--batch", "4000",
                  "--buffer-sets", "2", "--no-cpu-baseline", "--gather-chunks", "2")
    assert r.returncode == 0, r.stderr[-3000:]
    out = last_json(r.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 10 and out["scaling"] == "weak"
    assert out["value"] > 0 and out["roofline"]["frac"] > 0
    g = out["gather"]
    assert g["chunks"] == 2 and g["solve_plus_gather_ms"] > 0 and g["gathered_bytes_per_rank"] == 2 * 4000 * 8 * 3 * 10 * 8


@pytest.mark.gpu
def test_single_gpu_bench_line_has_the_contract_fields():
    r = run_bench("--steps", "20", "--warmup", "5", "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-3000:]
    out = last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out
    assert out["n_gpus"] == 1 and out["dtype"] == "f64" and out["config"]["buffer_sets"] == 16
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["traffic"] is None and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["bytes_per_launch"] == 10_000 * 2392
    assert "resident_buffers" in out["extra"]
    two = out["extra"]["two_streams_steady_state"]      # pipeline over two streams: reported beside, never as `value`
    assert two["launches"] == 2000 and 0 < two["us_per_launch"] < rf["kernel_us"] * 1.2
