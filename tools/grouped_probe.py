"""Grouped 3x3 (ResNeXt-101-64x4d, 512x768 x bs16) timings: this repo's kernel vs torch (MIOpen)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import ssad_amd  # noqa
from ssad_amd import kernels as K

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

N = 16
for name, C, H, W, s, cnt in (("res2", 256, 128, 192, 1, 3), ("res3.0", 512, 128, 192, 2, 1), ("res3", 512, 64, 96, 1, 3),
                              ("res4.0", 1024, 64, 96, 2, 1), ("res4", 1024, 32, 48, 1, 22), ("res5.0", 2048, 32, 48, 2, 1),
                              ("res5", 2048, 16, 24, 1, 2)):
    x = torch.randn(N, C, H, W, device="cuda"); w = torch.randn(C, C // 64, 3, 3, device="cuda") * 0.05
    b = torch.randn(C, device="cuda")
    y = torch.empty(N, C, (H - 1) // s + 1, (W - 1) // s + 1, device="cuda")
    pk = K.grouped_conv3x3_pack_filter(w, 64)
    ours = t(lambda: K.grouped_conv3x3_forward(x, None, b, 64, s, True, out=y, packed=pk))
    ref = t(lambda: F.conv2d(x, w, b, s, 1, 1, 64))
    fl = 2.0 * 9 * C * (C // 64) * y.shape[0] * y.shape[2] * y.shape[3]
    print("%-7s C=%4d cg=%2d %3dx%3d s%d x%2d: ours %.3f ms (%.1f TF/s)  torch %.3f ms" % (
        name, C, C // 64, H, W, s, cnt, ours, fl / ours / 1e9, ref))
