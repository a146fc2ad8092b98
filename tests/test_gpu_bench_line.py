"""bench.py's one JSON line, from a short run in a child process: the keys of the harness contract, the roofline objects and
their internal consistency. The numbers themselves are not judged here (tests/test_zy_perf_floor.py is the opt-in floor)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_contract():
    env = dict(os.environ, RT_BENCH_SPP_PER_STEP="8")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-secondary", "--no-c1", "--no-cpu-baseline"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "Mrays/s" and d["value"] > 0 and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = rays of the timed region / its time; a step is one pass of 8 spp here
    samples = 3 * 8 * 1920 * 1080
    assert abs(d["samples_per_sec"] * d["ms_per_step"] * 1e-3 * 3 - samples) <= 1e-6 * samples
    assert abs(d["value"] * 1e6 - d["rays_per_sample"] * d["samples_per_sec"]) <= 1e-6 * d["value"] * 1e6
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9
    v = r["valu_issue"]
    assert abs(v["frac"] - v["achieved"] / v["peak"]) <= 1e-9 and 0.05 < v["frac"] < 1.5
    s = d["roofline_shade"]
    assert s["bound"] == "valu-issue" and abs(s["frac"] - s["achieved"] / s["peak"]) <= 1e-9
    assert d["strong_1024spp"]["per_gpu_passes"] == 18 and d["strong_1024spp"]["seconds"] > 0
