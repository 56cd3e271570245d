#!/usr/bin/env python3
"""Golden learning-rate table: values of the REFERENCE's own
detectron/lib/utils/lr_policy.get_lr_at_iter, imported in the build container from /root/reference
(with make_head_graph.py's stubs; cfg.SOLVER fields set by assignment), for the distillation yaml's
solver section and for the other two policies.  Output: tests/golden/lr_table.json (data only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_lr_table.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_head_graph import REF, install_stubs   # noqa: E402

OUT = os.path.join(HERE, "lr_table.json")

CASES = {
    # configs/focal_distillation/retinanet_R-50-FPN_distillation.yaml:6-13
    "distillation_yaml": dict(BASE_LR=0.01, LR_POLICY="steps_with_decay", GAMMA=0.1, MAX_ITER=270000,
                              STEPS=[0, 180000, 240000], WARM_UP_ITERS=1000, WARM_UP_FACTOR=1.0 / 3.0,
                              WARM_UP_METHOD="linear"),
    "step_default": dict(BASE_LR=0.001, LR_POLICY="step", GAMMA=0.1, STEP_SIZE=30000, MAX_ITER=40000,
                         WARM_UP_ITERS=500, WARM_UP_FACTOR=1.0 / 3.0, WARM_UP_METHOD="linear"),
    "steps_with_lrs_constant_warmup": dict(LR_POLICY="steps_with_lrs", STEPS=[0, 60, 80], LRS=[0.02, 0.002, 0.0002],
                                           MAX_ITER=90, WARM_UP_ITERS=5, WARM_UP_FACTOR=0.25,
                                           WARM_UP_METHOD="constant"),
}
ITERS = {
    "distillation_yaml": list(range(0, 12)) + [250, 499, 500, 501, 998, 999, 1000, 1001, 90000, 179999, 180000,
                                               180001, 239999, 240000, 240001, 269999],
    "step_default": [0, 1, 100, 499, 500, 29999, 30000, 30001, 39999],
    "steps_with_lrs_constant_warmup": list(range(0, 8)) + [59, 60, 61, 79, 80, 89],
}


def main():
    install_stubs()
    sys.path.insert(0, REF)
    from core.config import cfg
    from utils import lr_policy
    table = {}
    for name, fields in CASES.items():
        for k, v in fields.items():
            setattr(cfg.SOLVER, k, v)
        rows = []
        for it in ITERS[name]:
            lr = lr_policy.get_lr_at_iter(it)
            rows.append({"iter": it, "lr": float(lr), "lr_f32_hex": float(lr).hex(), "dtype": str(lr.dtype)})
        table[name] = {"solver": {k: v for k, v in fields.items()}, "rows": rows}
    with open(OUT, "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print("wrote", OUT, {k: len(v["rows"]) for k, v in table.items()})


if __name__ == "__main__":
    main()
