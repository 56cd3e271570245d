"""bench.py's contract with the driver, exercised end to end on the GPU: `python bench.py --steps K --warmup W`
prints ONE JSON line (the last line of stdout) with the metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config keys, the `roofline` object of the
dominant kernel measured inside the timed region, and (round 4) BASELINE's other single-GPU workloads in `also`
without touching the headline number."""
import json
import os
import subprocess
import sys

import numpy as np
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
    # the dominant launch (subnet tower forward) runs on the split-operand engine by default: executed flops =
    # 3 x direct-form (exec_div 1/3) against the dense fp16 MFMA peak; on F(2x4) (SSAD_SPLIT_CONV=0): direct-form / 3
    # against the fp32 MFMA peak; F(2x2): / 2.25
    split = "split-operand" in r["kernel"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == (2500.0 if split else 157.3)
    assert 0.3 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    want_div = 1.0 / 3.0 if split else 3.0 if "F(2x4" in r["kernel"] else 2.25
    assert abs(r["exec_div"] - want_div) < 1e-9
    assert abs(r["achieved"] * 1e12 - r["flops_per_launch"] / r["exec_div"] / (r["avg_launch_ms"] * 1e-3)) <= 0.01 * r["achieved"] * 1e12
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
    # round 5: the drop-in route is timed in the same line, and `traffic` says which capture it comes from
    osf = also["operator_surface"]
    assert "error" not in osf, osf
    assert osf["lowered"]["finite"] and osf["as_written"]["finite"] and osf["batch_per_gpu"] == 16
    assert osf["lowered"]["operators_run"] < osf["lowered"]["operators_written"] == osf["as_written"]["operators_run"]
    assert osf["lowered"]["filter_packs_per_step"] == 20 and osf["as_written"]["filter_packs_per_step"] == 100
    assert osf["lowered"]["ms_per_step"] < osf["as_written"]["ms_per_step"]
    assert osf["lowered_over_program"] < 1.3, osf            # VERDICT r4 item 2's bar
    # round 6: the headline configuration with the split-operand engine off (fp32 MFMA instructions only) in the same
    # line, its dominant launch on F(2x4) against the fp32 MFMA peak; the headline's dtype note says what differs
    f32 = also["cfg3_fp32_mfma_only"]
    assert "error" not in f32 and f32["finite"] and f32["batch_per_gpu"] == 16 and f32["dtype"] == "f32", f32
    assert f32["roofline"]["peak"] == 157.3 and "F(2x4" in f32["roofline"]["kernel"]
    assert "split-operand" in d["dtype_note"] and d["dtype"] == "f32"
    assert np.allclose(osf["lowered"]["distill_loss"], osf["as_written"]["distill_loss"], rtol=1e-4)
    assert r.get("traffic_from_profile_round", "").startswith("r")
    assert d["config"]["collectives_per_step"] == 0 and d["config"]["parallelism"] == "dp1"


def test_scale_readiness_forced_one_rank_communicator():
    """the line a SCALE run produces, exercised on one GPU with the collectives forced onto a
    one-rank RCCL communicator (SSAD_DP_FORCE=1): n_gpus / parallelism present, bucket all-reduces really issued
    inside the timed steps (subnets: 2 buckets, backbone: 4), and a step that issues them costs what a step that
    does not costs (the GPU_MAX_HW_QUEUES=7 setting of bench.py) -- within 2 % here (1 % is the
    in-call repeat spread of the step itself)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"], env["MASTER_PORT"] = "127.0.0.1", "29531"
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3",
            "--profile-steps", "0", "--no-cpu-baseline", "--no-also"]
    out = {}
    for key, extra in (("plain", {}), ("dp", {"SSAD_DP_FORCE": "1", "RANK": "0", "LOCAL_RANK": "0",
                                              "WORLD_SIZE": "1"})):
        p = subprocess.run(base, env=dict(env, **extra), capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        out[key] = json.loads([l for l in p.stdout.splitlines() if l.startswith('{"metric"')][-1])
    dp, plain = out["dp"], out["plain"]
    assert dp["n_gpus"] == 1 and dp["config"]["parallelism"] == "dp1" and dp["scaling"] == "weak"
    assert plain["config"]["collectives_per_step"] == 0
    assert dp["config"]["collectives_per_step"] == 6, dp["config"]["collectives_per_step"]
    assert abs(dp["ms_per_step"] - plain["ms_per_step"]) <= 0.02 * plain["ms_per_step"], (dp["ms_per_step"],
                                                                                          plain["ms_per_step"])
    assert np.allclose(dp["config"]["distill_loss"], plain["config"]["distill_loss"], rtol=1e-6)
