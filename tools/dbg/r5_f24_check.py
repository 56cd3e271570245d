"""F(2x4, 3x3) engine: error against a float64 direct convolution and time against the F(2x2) engine, bs 16."""
import sys, time
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import ssad_amd
from ssad_amd import kernels as K

def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def check(N, ci, co, H, W, relu=True):
    g = torch.Generator(device="cuda").manual_seed(ci + co + H)
    x = torch.randn(N, ci, H, W, device="cuda", generator=g).clamp_(min=0)
    w = torch.randn(co, ci, 3, 3, device="cuda", generator=g) * float(1.0 / (3 * np.sqrt(ci)))
    b = torch.randn(co, device="cuda", generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    if relu: ref = ref.clamp_(min=0)
    p24 = K.conv_wino24_pack_filter(w)
    p22, _ = K.conv_wino_pack_filter(w, True, False)
    y24 = K.conv3x3_forward_wino24([x], p24, b, co, relu=relu)[0]
    y22 = K.conv3x3_forward([x], p22, b, co, relu=relu, wino=True)[0]
    sc = ref.abs().max().item()
    e24, e22 = (y24.double() - ref).abs().max().item() / sc, (y22.double() - ref).abs().max().item() / sc
    y24b = K.conv3x3_forward_wino24([x], p24, b, co, relu=relu)[0]
    out = [torch.empty_like(y24)]
    t24 = timeit(lambda: K.conv3x3_forward_wino24([x], p24, b, co, relu=relu, out=out))
    t22 = timeit(lambda: K.conv3x3_forward([x], p22, b, co, relu=relu, wino=True, out=out))
    print("N%d %4d->%3d @%3dx%3d: err/max F24 %.2e  F22 %.2e | reproducible %s | F24 %.3f ms  F22 %.3f ms  (%+.1f %%)"
          % (N, ci, co, H, W, e24, e22, bool(torch.equal(y24, y24b)), t24, t22, 100.0 * (t24 - t22) / t22), flush=True)

for shp in [(1, 16, 128, 8, 16), (1, 64, 128, 8, 16), (2, 256, 256, 10, 14), (2, 128, 128, 17, 33), (1, 256, 256, 5, 7), (2, 256, 720, 9, 13)]:
    check(*shp, relu=False)
for shp in [(16, 256, 256, 80, 112), (16, 256, 256, 40, 56), (16, 512, 512, 20, 28), (16, 128, 128, 80, 112), (16, 256, 720, 80, 112), (16, 256, 256, 20, 28)]:
    check(*shp)
