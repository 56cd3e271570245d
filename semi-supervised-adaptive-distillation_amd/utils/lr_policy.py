"""Learning-rate schedule of the training loop: detectron/lib/utils/lr_policy.py:28-123.

`get_lr_at_iter(it)` is what tools/train_net.py:172 evaluates before every iteration and hands to
`model.UpdateWorkspaceLr` (detector.py:594-648 = DistillHeads.update_lr / NativeResNetFPN.update_lr /
NativeDistillModel.update_lr here, momentum correction included).  The reference reads a global cfg;
here the SOLVER section is a dataclass whose defaults are core/config.py:577-642 and whose
`distillation()` constructor holds configs/focal_distillation/retinanet_R-50-FPN_distillation.yaml:6-13.

Pinned by tests/golden/lr_table.json: values of the reference's own get_lr_at_iter, written by
tests/golden/make_lr_table.py in the build container.
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np


@dataclass
class SolverConfig:
    base_lr: float = 0.001                  # config.py:580
    lr_policy: str = "step"                 # :584
    gamma: float = 0.1                      # :602
    step_size: int = 30000                  # :605
    steps: List[int] = field(default_factory=list)     # :609
    lrs: List[float] = field(default_factory=list)     # :612
    max_iter: int = 40000                   # :615
    momentum: float = 0.9                   # :618
    weight_decay: float = 0.0005            # :621
    warm_up_iters: int = 500                # :624
    warm_up_factor: float = 1.0 / 3.0       # :627
    warm_up_method: str = "linear"          # :630

    @classmethod
    def distillation(cls, num_gpus=8):
        """retinanet_R-50-FPN_distillation.yaml:6-13 (written for NUM_GPUS 8, bs 2 per GPU)."""
        return cls(base_lr=0.01, lr_policy="steps_with_decay", gamma=0.1, max_iter=270000,
                   steps=[0, 180000, 240000], weight_decay=0.0001, warm_up_iters=1000)


def get_step_index(solver, cur_iter):
    """lr_policy.py:107-114: which entry of STEPS the iteration falls in."""
    assert solver.steps[0] == 0, "The first step should always start at 0."
    steps = list(solver.steps) + [solver.max_iter]
    ind = 0
    for ind, step in enumerate(steps):
        if cur_iter < step:
            break
    return ind - 1


def _steps_with_lrs(solver, it):            # lr_policy.py:58-72
    return solver.lrs[get_step_index(solver, it)]


def _steps_with_decay(solver, it):          # lr_policy.py:75-91
    return solver.base_lr * solver.gamma ** get_step_index(solver, it)


def _step(solver, it):                      # lr_policy.py:94-99
    return solver.base_lr * solver.gamma ** (it // solver.step_size)


_POLICIES = {"steps_with_lrs": _steps_with_lrs, "steps_with_decay": _steps_with_decay, "step": _step}


def get_lr_at_iter(solver, it):
    """lr_policy.py:28-44: the policy's value, times the warm-up factor during the first
    WARM_UP_ITERS iterations (constant, or linear from WARM_UP_FACTOR to 1).  Returns np.float32
    like the reference: the comparison with the workspace's lr blob is exact (detector.py:602-604)."""
    if solver.lr_policy not in _POLICIES:
        raise NotImplementedError("Unknown LR policy: {}".format(solver.lr_policy))
    lr = _POLICIES[solver.lr_policy](solver, it)
    if it < solver.warm_up_iters:
        if solver.warm_up_method == "constant":
            warmup_factor = solver.warm_up_factor
        elif solver.warm_up_method == "linear":
            alpha = it / solver.warm_up_iters
            warmup_factor = solver.warm_up_factor * (1 - alpha) + alpha
        else:
            raise KeyError("Unknown SOLVER.WARM_UP_METHOD: {}".format(solver.warm_up_method))
        lr *= warmup_factor
    return np.float32(lr)


class LrSchedule(object):
    """train_net.py:171-173 for this build's models: `schedule.apply(model, it)` before `model.step(...)`."""

    def __init__(self, solver=None):
        self.solver = solver or SolverConfig.distillation()

    def __call__(self, it):
        return get_lr_at_iter(self.solver, it)

    def apply(self, model, it):
        """model: anything with update_lr(new_lr) (DistillHeads, NativeResNetFPN, NativeDistillModel)."""
        return model.update_lr(self(it))
