"""The bottleneck's last layer (Conv 1x1 -> folded AffineChannel -> Sum with the shortcut -> Relu) on its HBM-bound
shapes: gemm_conv.hip's GEMM with the fused epilogue (what the native backbones run) against conv1x1_fused.hip's
persistent kernel with W resident in LDS (round 1), and the first layer (bias + ReLU only).
    python tools/pw_tail_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssad_amd  # noqa
from ssad_amd import kernels as K


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


N = 16
for (ci, co, H, W, res) in ((64, 256, 160, 224, True), (64, 256, 160, 224, False), 
                            (128, 512, 80, 112, True), (512, 128, 80, 112, False), (256, 1024, 40, 56, True),
                            (64, 256, 128, 192, True)):
    x = torch.randn(N, ci, H, W, device="cuda")
    w = torch.randn(co, ci, 1, 1, device="cuda") * 0.05
    b = torch.randn(co, device="cuda")
    r = torch.randn(N, co, H, W, device="cuda") if res else None
    wt = K.transpose_filter(w)
    y = torch.empty(N, co, H, W, device="cuda")
    a = K.conv1x1_forward(x, wt, co, bias=b, residual=r, relu=True, out=y).clone()
    c = K.conv1x1_bias_act(x, w, b, r, relu=True)
    same = float((a - c).abs().max())
    t_g = timeit(lambda: K.conv1x1_forward(x, wt, co, bias=b, residual=r, relu=True, out=y))
    t_f = timeit(lambda: K.conv1x1_bias_act(x, w, b, r, relu=True))
    byts = 4.0 * N * H * W * (ci + co * (2 if res else 1))
    print("%4d->%4d @%3dx%3d res=%d  %.0f MB | gemm %.3f ms (%.2f TB/s) | fused %.3f ms (%.2f TB/s) | max diff %.1e" % (
        ci, co, H, W, res, byts / 1e6, t_g, byts / t_g / 1e9, t_f, byts / t_f / 1e9, same), flush=True)
