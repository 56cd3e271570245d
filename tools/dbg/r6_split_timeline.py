"""Cycle stamps of conv3x3_split_kernel (SPLIT_TIMELINE build): per item of workgroup 5, wave 0 -- main loop, hand-over
arithmetic, prologue issue, epilogue, landing wait, barrier.  One tower filter, 5 levels, bs 16."""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
import ssad_amd
from ssad_amd import kernels as K
g = torch.Generator(device="cuda").manual_seed(3)
N = 16
lv = [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)]
xs = [torch.randn(N, 256, h, w, device="cuda", generator=g) for h, w in lv] * 4
wt = torch.randn(256, 256, 3, 3, device="cuda", generator=g) * 0.02
pf = K.conv_split_pack_filter(wt)
for _ in range(3):
    K.conv3x3_forward_split(xs, pf, None, 256, relu=True)
torch.cuda.synchronize()
buf = np.zeros((64, 8), np.uint64)
raw = ctypes.CDLL(K.LIB_PATH)
assert raw.ssad_split_dbg_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
names = ["main loop", "drain + decode", "bind + prologue issue", "epilogue", "landing wait", "barrier"]
rows = []
for i in range(64):
    st = buf[i].astype(np.int64)
    if st[6] == 0: break
    rows.append([st[k + 1] - st[k] for k in range(6)] + [st[6] - st[0]])
rows = np.array(rows[1:-1], np.float64)          # s_memtime ticks (100 MHz) -> shown as is
print("items", len(rows), "(units: s_memtime ticks)")
for k, nm in enumerate(names + ["item total"]):
    print("%-24s mean %8.1f  min %8.1f  max %8.1f" % (nm, rows[:, k].mean(), rows[:, k].min(), rows[:, k].max()))
