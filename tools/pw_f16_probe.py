#!/usr/bin/env python3
"""Isolated timings of the fp16 pointwise GEMM (gemm_f16.hip) on the backbone shapes of BASELINE
config 5 (512 x 768, bs 16): achieved algorithmic GB/s (x + y (+ residual) + filter, fp16) and
TFLOP/s per shape.  Development aid."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssad_amd  # noqa
from ssad_amd import kernels as K


def timeit(fn, iters=30, warm=5):
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


def main():
    L = K.lib()
    N = 16
    st = torch.cuda.current_stream().cuda_stream
    shapes = [("res2 c1 256->64", 256, 64, 128, 192, False), ("res2 c3 64->256 +res", 64, 256, 128, 192, True),
              ("x101 res2 c1 256->256", 256, 256, 128, 192, False),
              ("res3 c1 512->128", 512, 128, 64, 96, False), ("res3 c3 128->512 +res", 128, 512, 64, 96, True),
              ("x101 res3 512->512", 512, 512, 64, 96, True),
              ("res4 c1 1024->256", 1024, 256, 32, 48, False), ("res4 c3 256->1024 +res", 256, 1024, 32, 48, True),
              ("x101 res4 1024->1024", 1024, 1024, 32, 48, True),
              ("res5 c1 2048->512", 2048, 512, 16, 24, False), ("res5 c3 512->2048 +res", 512, 2048, 16, 24, True),
              ("x101 res5 2048->2048", 2048, 2048, 16, 24, True),
              ("lat res3 512->256 +up", 512, 256, 64, 96, False)]
    tot = 0.0
    only = sys.argv[1] if len(sys.argv) > 1 else ""          # substring filter on the shape's name
    for name, Cc, M, H, W, res in shapes:
        if only not in name:
            continue
        x = torch.randn((N, Cc // 8, H, W, 8), device="cuda").half()
        y = torch.empty((N, M // 8, H, W, 8), device="cuda", dtype=torch.float16)
        r = torch.randn((N, M // 8, H, W, 8), device="cuda").half() if res else None
        w = torch.randn((M, Cc), device="cuda") * 0.05
        b = torch.zeros(M, device="cuda")
        wf = torch.empty(L.ssad_pw_f16_filter_halves(M, Cc), dtype=torch.float16, device="cuda")
        L.ssad_pw_f16_pack_filter(w.data_ptr(), M, Cc, wf.data_ptr(), None, st)
        d = K.PwF16()
        d.x, d.w, d.y, d.bias = x.data_ptr(), wf.data_ptr(), y.data_ptr(), b.data_ptr()
        d.residual = r.data_ptr() if res else None
        d.N, d.C, d.M, d.Ho, d.Wo, d.Hi, d.Wi, d.stride, d.flags = N, Cc, M, H, W, H, W, 1, K.CONV_RELU
        t = timeit(lambda: L.ssad_conv1x1_f16(C.byref(d), st))
        px = N * H * W
        byts = 2.0 * px * (Cc + M * (2 if res else 1)) + 2.0 * M * Cc
        fl = 2.0 * px * Cc * M
        tot += t
        print("%-26s %7.3f ms  %7.1f GB/s  %7.1f TF/s  (min-time: hbm %.3f ms @6.4TB/s, mfma %.3f ms @2.5PF)" % (
            name, t, byts / t / 1e6, fl / t / 1e9, byts / 6.4e9, fl / 2.5e12))
    print("sum %.3f ms" % tot)


if __name__ == "__main__":
    main()
