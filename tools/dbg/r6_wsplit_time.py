"""Isolated launch times of the split-operand filter gradient against the F(3x3, 2x2) engine on config 3's shapes."""
import os, sys, time
import torch
sys.path.insert(0, ".")
import ssad_amd
from ssad_amd import kernels as K


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(name, N, C, M, shapes):
    gen = torch.Generator(device="cuda").manual_seed(1)
    Xs = [torch.randn((N, C, h, w), device="cuda", generator=gen) for h, w in shapes]
    dYs = [torch.randn((N, M, h, w), device="cuda", generator=gen) * 1e-3 for h, w in shapes]
    px = sum(N * h * w for h, w in shapes)
    fl = 2.0 * 9 * C * M * px
    only = os.environ.get("WSPLIT_ONLY") == "1"     # the split engine alone, without the bias gradient
    if only:
        ts = timeit(lambda: K.conv3x3_wgrad(Xs, dYs, M, split=True, want_db=False))
        print("%-28s split %.3f ms (%.0f TF/s direct-form)" % (name, ts, fl / ts / 1e9), flush=True)
        return
    tw = timeit(lambda: K.conv3x3_wgrad(Xs, dYs, M))
    ts = timeit(lambda: K.conv3x3_wgrad(Xs, dYs, M, split=True))
    a = K.conv3x3_wgrad(Xs, dYs, M)[0].clone()
    b = K.conv3x3_wgrad(Xs, dYs, M, split=True)[0]
    err = ((a - b).abs().max() / a.abs().max()).item()
    print("%-28s winograd %.3f ms (%.0f TF/s direct-form)   split %.3f ms (%.0f TF/s)   max diff / max %.2e"
          % (name, tw, fl / tw / 1e9, ts, fl / ts / 1e9, err), flush=True)


fpn = [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)]
case("tower 256->256, 5 levels", 16, 256, 256, fpn)
case("cls_pred 256->720", 16, 256, 720, fpn)
case("res3 128->128 80x112", 16, 128, 128, [(80, 112)])
case("res4 256->256 40x56", 16, 256, 256, [(40, 56)])
case("res5 512->512 20x28", 16, 512, 512, [(20, 28)])
