"""Launch-geometry sweep of the fused classification-loss kernel at bs 16 (run once per setting of
SSAD_LOSS_MAXBLOCKS / SSAD_LOSS_CGROUPS, which the library reads at first use)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssad_amd  # noqa: E402,F401
from ssad_amd import kernels as K, synth  # noqa: E402

N, A, C = 16, 9, 80
dev = "cuda"
lv = []
for h, w in synth.LEVEL_SHAPES_600:
    x = torch.randn((N, A * C, h, w), device=dev) * 2 - 4
    q = torch.sigmoid(torch.randn((N, A * C, h, w), device=dev) * 2 - 4).clamp_(1e-6, 1 - 1e-6)
    g = torch.where(torch.rand((N, A, h, w), device=dev) < 0.05, -1, 0).to(torch.int32)
    lv.append((x, q, g))
E = sum(x.numel() for x, _, _ in lv)
norm = K.pow_sum([q for _, q, _ in lv], 1.8).reshape(1)
fg = torch.tensor([1000.0], device=dev)
outs = [torch.empty_like(x) for x, _, _ in lv]
dkw = dict(gamma=2.0, alpha=0.5, beta=0.0, num_classes=C, ignored_label=-1, scale=1.0)
fkw = dict(gamma=2.0, alpha=0.25, num_classes=C, scale=1.0)


def timeit(fn, iters=30, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


t = timeit(lambda: K.cls_losses_fused(lv, norm, fg, dkw, fkw, out=outs))
print("MAXBLOCKS=%s CGROUPS=%s fused %.4f ms %.1f GB/s (%.3f of 8 TB/s)" % (
    os.environ.get("SSAD_LOSS_MAXBLOCKS", "-"), os.environ.get("SSAD_LOSS_CGROUPS", "-"), t,
    E * 12.05 / t / 1e6, E * 12.05 / t / 1e6 / 8000))
t = timeit(lambda: K.pow_sum([q for _, q, _ in lv], 1.8))
print("   powsum %.4f ms %.1f GB/s" % (t, E * 4 / t / 1e6))
