"""bench.py's contract with the driver, exercised end to end on the GPU: `python bench.py --steps K --warmup W`
prints ONE JSON line (the last line of stdout) with the metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config keys, the `roofline` object of the
dominant kernel measured inside the timed region, and (round 4) BASELINE's other single-GPU workloads in `also`
without touching the headline number."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_and_also_legs():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                        "--profile-steps", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])
    assert sum(1 for l in lines if l.startswith('{"metric"')) == 1
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "kernels"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "images/s"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["metric"].startswith("images/sec (student+teacher fwd + student bwd) R50-FPN distill")
    assert abs(d["value"] - 16 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-2 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert 0.3 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] * 1e12 - r["flops_per_launch"] / 2.25 / (r["avg_launch_ms"] * 1e-3)) <= 0.01 * r["achieved"] * 1e12
    for key in ("roofline_loss", "roofline_pow_sum"):
        assert d[key]["bound"] == "hbm" and d[key]["peak"] == 8000.0 and 0.1 < d[key]["frac"] < 1.0
    also = d["also"]
    c5, c2 = also["cfg5_f16"], also["cfg2"]
    assert "error" not in c5 and "error" not in c2, (c5, c2)
    assert c5["finite"] and c5["batch_per_gpu"] == 16 and c5["image"] == "3x512x768" and c5["dtype"].startswith("f16")
    assert c5["roofline"]["peak"] == 2500.0 and 0.05 < c5["roofline"]["frac"] < 1.0
    assert abs(c5["images_per_s"] - 16 / (c5["ms_per_step"] * 1e-3)) <= 1e-2 * c5["images_per_s"]
    assert c2["finite"] and c2["batch_per_gpu"] == 2 and c2["image"] == "3x640x896" and c2["dtype"] == "f32"
    assert "X-101-64X4D" in c5["workload"] and "student only" in c2["workload"]
