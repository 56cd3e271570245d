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
for s in range(5, 9):
    r = [int(buf[s, k]) - int(buf[s, 0]) for k in (0, 5, 6, 1, 2)]
    print("FINE unit %d: vmcnt wait %d, barrier %d, setup_unit %d, steps %d; to next top %d" %
          (s, r[1], r[2] - r[1], r[3] - r[2], r[4] - r[3], int(buf[s + 1, 0]) - int(buf[s, 2])))
tot = [int(buf[s + 1, 0]) - int(buf[s, 0]) for s in range(4, 40)]
print("SUMMARY cycles/unit median %d min %d max %d" % (int(np.median(tot)), min(tot), max(tot)))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(10):
    K.conv3x3_wgrad(Xs, dYs, M, want_db=False)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / 10
fl = 2.0 * M * Cin * 9 * N * sum(h * w for h, w in shapes) / 2.25
print("SUMMARY wgrad all levels %.3f ms  %.1f TF/s executed" % (ms, fl / ms / 1e9))
