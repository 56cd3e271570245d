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
wv = np.zeros((8, 64, 2), dtype=np.uint64)
if hasattr(K.lib(), "ssad_dbg_read_waves") and K.lib().ssad_dbg_read_waves(wv.ctypes.data_as(C.c_void_p)) == 0:
    for s in range(20, 28):
        rel = int(wv[:, s - 1, 1].max())      # release of the previous chunk's barrier
        print("WAVES chunk %d: steps done at +%s, released +%d" %
              (s, [int(wv[w, s, 0]) - rel for w in range(8)], int(wv[0, s, 1]) - rel))
for s in range(3, 22):
    print("DMA block of chunk %d: %d cycles (tile decode every 16th)" % (s, int(buf[1, s, 7]) - int(buf[1, s, 6])))
tot = [int(buf[1, s + 1, 0]) - int(buf[1, s, 0]) for s in range(2, 14)]
print("SUMMARY cycles/chunk median %d min %d max %d (chunks 2-13 of the first tile)" % (int(np.median(tot)), min(tot), max(tot)))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(10):
    K.conv3x3_forward(Xs, wf, b, M, relu=True, out=Ys, wino=True)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / 10
fl = 2.0 * M * Cin * 9 * N * sum(h * w for h, w in shapes) / 2.25
print("SUMMARY forward all levels %.3f ms  %.1f TF/s executed = %.3f of 157.3" % (ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3))
