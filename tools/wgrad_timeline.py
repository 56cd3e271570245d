#!/usr/bin/env python3
"""Debug aid: per-unit cycle stamps of wave 0 of one workgroup of the Winograd
weight-gradient kernel.  Needs conv3x3_wgrad_winograd.hip compiled with
-DWGRAD_TIMELINE into a throw-away copy of the library."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssad_amd  # noqa
from ssad_amd import kernels as K, synth

N, M, Cin = 16, 256, 256
shapes = synth.LEVEL_SHAPES_600
Xs = [torch.randn((N, Cin, h, w), device="cuda") for h, w in shapes]
dYs = [torch.randn((N, M, h, w), device="cuda") for h, w in shapes]
for _ in range(5):
    K.conv3x3_wgrad(Xs, dYs, M, want_db=False)
torch.cuda.synchronize()
buf = np.zeros((64, 8), dtype=np.uint64)
rc = K.lib().ssad_wdbg_read(buf.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
t0 = int(buf[0, 0])
print("unit: [start, +setup, +4 k-steps, +stores, +barrier]")
for s in range(40):
    r = [int(v) - t0 for v in buf[s, :5]]
    print(s, r, "setup=%d steps=%d store=%d barrier=%d total=%d" %
          (r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[4] - r[0]))
