"""bench.py end to end on the GPU (small step counts): the one JSON line the driver parses, with the `roofline` and
`cpu_baseline` objects, for the headline env and for the humanoid."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_headline_line_contract():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    d = _run("--steps", "60", "--warmup", "10", "--batch", "4096", "--cpu-sample-steps", "2")
    assert d["metric"] == base["metric"] and d["unit"] == "env-steps/s"
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["warmup"] == 10 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "RandomHopper-v0" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] - 4096 * 60 / (d["ms_per_step"] * 60 / 1e3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert abs(r["achieved"] - 173 * 4096 / (r["kernel_avg_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]   # algorithmic bytes / launch time
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["cores"] >= 1 and c["value"] > 0 and c["one_env_one_core"] > 0
    assert d["nonfinite_lanes"] == 0 and d["solver_capped_waves"] == 0


def test_humanoid_line():
    d = _run("--env", "RandomHumanoid-v0", "--steps", "12", "--warmup", "3", "--batch", "2048", "--no-cpu-baseline")
    assert "RandomHumanoid-v0" in d["config"]["workload"] and d["value"] > 1e5
    assert d["roofline"]["bytes_per_env_step"] == 2073 and d["roofline"]["kernel"] == "humanoid_step_kernel"
    assert d["nonfinite_lanes"] == 0
