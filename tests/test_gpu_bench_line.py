"""The line the driver records: `python bench.py --gpus 1 --steps 20 --warmup 5` prints ONE JSON object with the contract's keys, a
roofline of the warp kernel measured in the same process, and the steps from the idle device beside the preheated ones."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_in_the_drivers_form():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline", "--no-host-paths", "--no-c4"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "one JSON line on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "preheat_s", "from_idle"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["unit"] == "panoramas/s" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value is K steps over the timed region
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 1e-3
    assert d["from_idle"]["value"] > 0 and d["preheat_s"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # algorithmic bytes over the measured mean launch duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / r["avg_launch_us"] / 1e3) < 0.01 * r["achieved"]
    assert 0.2 < r["frac"] < 0.7 and 0.2 < r["cold"]["frac"] <= r["frac"] + 0.02
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"]
