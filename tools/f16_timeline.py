"""Debug aid: where a workgroup of the fp16 forward kernel spends its time (s_memtime stamps of
wave 0: entry, main loop start, main loop end, exit).  Needs the library rebuilt with
`make -C semi-supervised-adaptive-distillation_amd/csrc EXTRA=-DF16_TIMELINE` (never ship that build)."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssad_amd  # noqa
from ssad_amd import kernels as K

N, ci, co, H, W = 16, 256, 256, 80, 112
x = torch.randn(N, ci, H, W, device="cuda")
w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
b = torch.randn(co, device="cuda")
xb = K.f16_pack_activations(x)
w16, _ = K.f16_pack_filter(w, True, False)
for _ in range(3):
    K.conv3x3_forward_f16(xb, w16, b, ci, co, relu=True)
torch.cuda.synchronize()
buf = np.zeros((4096, 4), dtype=np.uint64)
L = K.lib()
L.ssad_f16_dbg_read.argtypes = [C.c_void_p]
assert L.ssad_f16_dbg_read(buf.ctypes.data_as(C.c_void_p)) == 0
wgs = N * 35 * 2
s = buf[:min(wgs, 4096)].astype(np.int64)
t0 = s[:, 0].min()
span = s[:, 3].max() - t0
pro, loop, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
print("workgroups %d, kernel span %d ticks" % (wgs, span))
for name, v in (("prologue", pro), ("main loop", loop), ("epilogue", epi)):
    print("%-10s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f  (%.1f %% of a workgroup)" % (
        name, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90),
        100.0 * v.mean() / (pro + loop + epi).mean()))
start = np.sort(s[:, 0] - t0)
print("workgroup start times: first 512 by %d, 1024th at %d, last at %d" % (start[min(511, len(start) - 1)],
      start[min(1023, len(start) - 1)], start[-1]))
order = np.argsort(s[:, 0])
print("first 6 workgroups (start, prologue, loop, epilogue):")
for i in order[:6]:
    print("  wg %4d  %8d %6d %6d %6d" % (i, s[i, 0] - t0, pro[i], loop[i], epi[i]))
print("last 6:")
for i in order[-6:]:
    print("  wg %4d  %8d %6d %6d %6d" % (i, s[i, 0] - t0, pro[i], loop[i], epi[i]))
