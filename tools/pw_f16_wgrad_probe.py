#!/usr/bin/env python3
"""Isolated timings of the fp16 pointwise FILTER GRADIENT (conv3x3_wgrad_f16_kernel<true> + reduce + bias) on the
backbone shapes of BASELINE config 5 (512 x 768, bs 16).  Development aid."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssad_amd  # noqa
from ssad_amd import kernels as K
from pw_f16_probe import timeit

L = K.lib()
N = 16
st = torch.cuda.current_stream().cuda_stream
tot = 0.0
for name, Cc, M, H, W, cnt in (("res2 c1 256->64", 256, 64, 128, 192, 3), ("res2 c3 64->256", 64, 256, 128, 192, 3),
                               ("res3 c1 512->128", 512, 128, 64, 96, 4), ("res3 c3 128->512", 128, 512, 64, 96, 4),
                               ("res4 c1 1024->256", 1024, 256, 32, 48, 23), ("res4 c3 256->1024", 256, 1024, 32, 48, 23),
                               ("res5 c1 2048->512", 2048, 512, 16, 24, 3), ("res5 c3 512->2048", 512, 2048, 16, 24, 3),
                               ("lat 1024->256", 1024, 256, 32, 48, 1)):
    x = torch.randn((N, Cc // 8, H, W, 8), device="cuda").half()
    dy = torch.randn((N, M // 8, H, W, 8), device="cuda").half()
    nb = L.ssad_conv1x1_wgrad_f16_workspace_bytes(N, Cc, H, W, M)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    dw = torch.empty((M, Cc), device="cuda")
    inv = torch.ones(1, device="cuda")
    t = timeit(lambda: L.ssad_conv1x1_wgrad_f16(x.data_ptr(), dy.data_ptr(), N, Cc, H, W, M, 0, 1.0, inv.data_ptr(),
                                                dw.data_ptr(), None, ws.data_ptr(), nb, st))
    px = N * H * W
    byts = 2.0 * px * (Cc + M) + 4.0 * M * Cc
    fl = 2.0 * px * Cc * M
    tot += t * cnt
    print("%-20s %7.3f ms x %2d  %7.1f GB/s %7.1f TF/s  ws %.1f MB (min-time: hbm %.3f ms @6.4TB/s, mfma %.3f @2.5PF)" % (
        name, t, cnt, byts / t / 1e6, fl / t / 1e9, nb / 1e6, byts / 6.4e9, fl / 2.5e12), flush=True)
print("R-101 student, all pointwise filter gradients of res3..res5 (+ listed): %.2f ms" % tot)
