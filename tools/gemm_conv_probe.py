"""Pointwise-convolution probe: this repo's fp32-MFMA GEMM kernels (gemm_conv.hip) against torch's
BLAS front end (rocBLAS / hipBLASLt strided-batched GEMM, with the committed TunableOp picks when
SSAD_TUNED=1) on the 1x1 geometries of the ResNet bottlenecks at 640x896, batch 16.

    python tools/gemm_conv_probe.py [--batch 16]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssad_amd  # noqa: E402,F401
from ssad_amd import kernels as K  # noqa: E402

SHAPES = [  # (Cin, Cout, H, W)
    (64, 64, 160, 224), (64, 256, 160, 224), (256, 64, 160, 224),
    (256, 128, 80, 112), (128, 512, 80, 112), (512, 128, 80, 112), (256, 512, 80, 112),
    (512, 256, 40, 56), (256, 1024, 40, 56), (1024, 256, 40, 56), (512, 1024, 40, 56),
    (1024, 512, 20, 28), (512, 2048, 20, 28), (2048, 512, 20, 28), (1024, 2048, 20, 28),
    (2048, 256, 20, 28), (1024, 256, 40, 56), (512, 256, 80, 112),
]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    if os.environ.get("SSAD_TUNED") == "1":
        from tools.harness import full_model
        full_model.setup_tunableop()
    N = a.batch
    tot = [0.0] * 6
    for (ci, co, H, W) in SHAPES:
        x = torch.randn(N, ci, H, W, device="cuda")
        w = torch.randn(co, ci, 1, 1, device="cuda") * 0.05
        dy = torch.randn(N, co, H, W, device="cuda")
        w2 = w.view(co, ci)
        P = H * W
        wt = K.transpose_filter(w)
        y = torch.empty(N, co, H, W, device="cuda")
        dx = torch.empty_like(x)
        dw = torch.empty(co, ci, device="cuda")
        f_me = timeit(lambda: K.conv1x1_forward(x, wt, co, out=y))
        f_mm = timeit(lambda: torch.bmm(w2.view(1, co, ci).expand(N, co, ci), x.view(N, ci, P), out=y.view(N, co, P)))
        d_me = timeit(lambda: K.conv1x1_dgrad(dy, w, accumulate_into=None))
        d_mm = timeit(lambda: torch.bmm(w2.t().reshape(1, ci, co).expand(N, ci, co), dy.view(N, co, P),
                                        out=dx.view(N, ci, P)))
        g_me = timeit(lambda: K.conv1x1_wgrad(x, dy, out=dw))
        g_mm = timeit(lambda: torch.bmm(dy.view(N, co, P), x.view(N, ci, P).transpose(1, 2)).sum(0))
        fl = 2.0 * N * P * ci * co / 1e9
        print("%4d->%4d @%3dx%3d %6.1f GF | fwd ours %.3f (%3.0f TF) blas %.3f (%3.0f) | dgrad ours %.3f (%3.0f) "
              "blas %.3f (%3.0f) | wgrad ours %.3f (%3.0f) blas %.3f (%3.0f)" % (
                  ci, co, H, W, fl, f_me, fl / f_me, f_mm, fl / f_mm, d_me, fl / d_me, d_mm, fl / d_mm,
                  g_me, fl / g_me, g_mm, fl / g_mm), flush=True)
        for i, v in enumerate((f_me, f_mm, d_me, d_mm, g_me, g_mm)):
            tot[i] += v
    print("sum ms: fwd ours %.2f blas %.2f | dgrad ours %.2f blas %.2f | wgrad ours %.2f blas %.2f" % tuple(tot))


if __name__ == "__main__":
    main()
