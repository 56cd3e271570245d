#!/usr/bin/env python3
"""Debug aid: per-chunk cycle stamps (s_memtime) of wave 0 of one workgroup of the
persistent Winograd kernel.  Needs conv3x3_winograd.hip compiled with
-DWINO_TIMELINE into a throw-away copy of the library (never ship that build:
it writes a debug buffer from the hot loop)."""
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
Wt = torch.randn((M, Cin, 3, 3), device="cuda") * 0.01
b = torch.zeros(M, device="cuda")
Ys = [torch.empty((N, M, h, w), device="cuda") for h, w in shapes]
wf, wd = K.conv_wino_pack_filter(Wt)
for _ in range(5):
    K.conv3x3_forward(Xs, wf, b, M, relu=True, out=Ys, wino=True)
torch.cuda.synchronize()
buf = np.zeros((2, 64, 8), dtype=np.uint64)
rc = K.lib().ssad_dbg_read(buf.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
t0 = int(buf[0, 0, 0])
print("wave 0 of workgroup 3: chunk, [start, +setup, +mfma steps, +barrier, epilogue start, end]")
for s in range(40):
    r = [int(v) - t0 if v else 0 for v in buf[1, s, :6]]
    print(s, r, "setup=%d steps=%d wait=%d" % (r[1] - r[0], r[2] - r[1], r[3] - r[2]),
          ("epilogue=%d" % (r[5] - r[4])) if r[5] else "")
