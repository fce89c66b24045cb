"""bench.py prints ONE JSON line with the driver's contract (run at a tiny size so it takes seconds)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.gpu
@pytest.mark.parametrize("renderer", ["trilinear", "siddon"])
def test_bench_json_contract(renderer):
    cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--size", "64", "--det", "32",
           "--batch", "4", "--n-points", "80", "--renderer", renderer, "--cpu-rays", "256"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 4 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"] + 1e-9
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["units_per_launch"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["avg_launch_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "DRRs/s" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
