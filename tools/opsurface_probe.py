"""Time the subnets' training iteration on the operator surface (operator_surface.HeadsNetStep: CreateNet once,
one RunNet per net and iteration) at bs 16 / 600 px, lowered and as written, beside the hand-built program."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ssad_amd  # noqa: F401
from ssad_amd import synth
from ssad_amd.caffe2_hip import dyndep, workspace
from ssad_amd.operator_surface import HeadsNetStep


def run(lowering, N, steps, warmup, update=True):
    workspace.ResetWorkspace()
    st = HeadsNetStep(N=N, shapes=synth.LEVEL_SHAPES_600, update=update, lowering=lowering)
    st.feed_params()
    st.feed_synthetic()
    st.create()
    for _ in range(warmup):
        st.step()
    torch.cuda.synchronize()
    p0, c0 = workspace.Counter("filter_packs"), workspace.Counter("conv_launch_calls")
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    low = st.lowered()
    print("lowering=%d: %.2f ms/step, %d ops run (of %d written), %.1f filter packs and %.1f conv launcher calls per step, losses %s"
          % (lowering, dt, len(low["teacher"]) + len(low["student"]), st.total_ops,
             (workspace.Counter("filter_packs") - p0) / steps, (workspace.Counter("conv_launch_calls") - c0) / steps,
             [round(v, 5) for k, v in sorted(st.losses().items()) if k.startswith("fl_distill")]), flush=True)
    workspace.ResetWorkspace()
    return dt


if __name__ == "__main__":
    dyndep.InitOpsLibrary()
    N = int(os.environ.get("N", "16"))
    steps, warmup = int(os.environ.get("STEPS", "5")), 2
    for low in ([1, 0] if os.environ.get("BOTH", "1") == "1" else [1]):
        run(low, N, steps, warmup)
