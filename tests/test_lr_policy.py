"""utils/lr_policy.py against the table the reference's own lr_policy.get_lr_at_iter produced
(tests/golden/lr_table.json, written by tests/golden/make_lr_table.py from /root/reference in the build
container): bit for bit in float32, as the reference compares the value with the workspace's lr blob exactly
(detectron/lib/modeling/detector.py:602-604)."""
import json
import os

import numpy as np
import pytest

import ssad_amd  # noqa: F401
from ssad_amd.utils import lr_policy as LP

KEYS = {"BASE_LR": "base_lr", "LR_POLICY": "lr_policy", "GAMMA": "gamma", "STEP_SIZE": "step_size",
        "STEPS": "steps", "LRS": "lrs", "MAX_ITER": "max_iter", "WARM_UP_ITERS": "warm_up_iters",
        "WARM_UP_FACTOR": "warm_up_factor", "WARM_UP_METHOD": "warm_up_method"}


@pytest.fixture(scope="module")
def table(golden_dir):
    with open(os.path.join(golden_dir, "lr_table.json")) as f:
        return json.load(f)


def test_lr_table_bit_for_bit(table):
    assert set(table) == {"distillation_yaml", "step_default", "steps_with_lrs_constant_warmup"}
    for name, case in table.items():
        solver = LP.SolverConfig(**{KEYS[k]: v for k, v in case["solver"].items()})
        for row in case["rows"]:
            lr = LP.get_lr_at_iter(solver, row["iter"])
            assert isinstance(lr, np.float32) and row["dtype"] == "float32"
            assert float(lr).hex() == row["lr_f32_hex"], (name, row["iter"], float(lr), row["lr"])


def test_distillation_yaml_is_the_default_schedule(table):
    sched = LP.LrSchedule()
    for row in table["distillation_yaml"]["rows"]:
        assert float(sched(row["iter"])).hex() == row["lr_f32_hex"]
    assert sched.solver.weight_decay == 0.0001 and sched.solver.momentum == 0.9
    # warm-up ends exactly at the base rate; the two decays are at 180k and 240k
    assert sched(1000) == np.float32(0.01) and sched(179999) == np.float32(0.01)
    assert sched(180000) == np.float32(0.01 * 0.1) and sched(240000) == np.float32(0.01 * 0.1 ** 2)


def test_errors_follow_the_reference():
    with pytest.raises(NotImplementedError, match="Unknown LR policy"):
        LP.get_lr_at_iter(LP.SolverConfig(lr_policy="cosine"), 0)
    with pytest.raises(KeyError, match="WARM_UP_METHOD"):
        LP.get_lr_at_iter(LP.SolverConfig(warm_up_method="exp"), 0)
    with pytest.raises(AssertionError):
        LP.get_lr_at_iter(LP.SolverConfig(lr_policy="steps_with_decay", steps=[10, 20]), 0)


def test_schedule_drives_update_lr():
    class Model(object):
        def __init__(self):
            self.seen = []

        def update_lr(self, lr):
            self.seen.append(lr)
            return lr
    m, sched = Model(), LP.LrSchedule()
    for it in (0, 1, 1000):
        assert sched.apply(m, it) == sched(it)
    assert [float(v) for v in m.seen] == [float(sched(0)), float(sched(1)), float(np.float32(0.01))]
