"""wino_wgrad_kernel (+ reduce + bias) at config 3's subnet shapes, bs 16: tower layer (256 -> 256, 5 levels), cls_pred."""
import sys, time
import torch
sys.path.insert(0, ".")
import ssad_amd
from ssad_amd import kernels as K

def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

g = torch.Generator(device="cuda").manual_seed(3)
N = 16
lv = [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)]
for name, ci, co in (("tower 256->256", 256, 256), ("cls_pred 256->720", 256, 720)):
    xs = [torch.randn(N, ci, h, w, device="cuda", generator=g) for h, w in lv]
    dys = [torch.randn(N, co, h, w, device="cuda", generator=g) for h, w in lv]
    dW = torch.empty((co, ci, 3, 3), device="cuda"); db = torch.empty(co, device="cuda")
    fn = lambda: K.conv3x3_wgrad(xs, dys, co, dW=dW, db=db)
    ms = timeit(fn)
    fl = 2.0 * 9 * ci * co * N * sum(h * w for h, w in lv)
    print("%-20s %8.3f ms  %6.1f TF/s direct-equivalent" % (name, ms, fl / ms / 1e9), flush=True)
