"""Isolated timing of the pointwise engines on the backbones' bottleneck shapes (bs 16, 640 x 896): the split-operand
GEMM (gemm_split.hip: |max| pass + filter split inside the timed region) against the exact-fp32 MFMA GEMM."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssad_amd  # noqa: E402,F401
from ssad_amd import kernels as K  # noqa: E402
import ctypes as C  # noqa: E402


def timeit(fn, n=10, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    L = K.lib()
    gen = torch.Generator(device="cuda").manual_seed(1)
    N = 16
    cases = [("res2 64->256 160x224", 64, 256, 160, 224), ("res2 256->64", 256, 64, 160, 224),
             ("res3 512->128 80x112", 512, 128, 80, 112), ("res3 128->512", 128, 512, 80, 112),
             ("res4 1024->256 40x56", 1024, 256, 40, 56), ("res4 256->1024", 256, 1024, 40, 56),
             ("res5 2048->512 20x28", 2048, 512, 20, 28), ("res5 512->2048", 512, 2048, 20, 28),
             ("fpn lateral 2048->256 20x28", 2048, 256, 20, 28), ("fpn lateral 1024->256 40x56", 1024, 256, 40, 56)]
    for name, Cin, M, H, W in cases:
        X = torch.randn((N, Cin, H, W), device="cuda", generator=gen).clamp_(min=0)
        Wt = torch.randn((M, Cin, 1, 1), device="cuda", generator=gen) * 0.03
        b = torch.randn(M, device="cuda", generator=gen)
        R = torch.randn((N, M, H, W), device="cuda", generator=gen)
        wt = K.transpose_filter(Wt)
        y = torch.empty((N, M, H, W), device="cuda")
        d = K.gemm_conv_desc(wt, wt.shape[1], X, y, Cin, M, b, R, None, True, False)
        ws = torch.empty(L.ssad_conv1x1_gemm_split_workspace_bytes(C.byref(d)), dtype=torch.uint8, device="cuda")
        t_f = timeit(lambda: K._check(L.ssad_conv1x1_gemm(C.byref(d), K._stream()), "gemm"))
        t_s = timeit(lambda: K._check(L.ssad_conv1x1_gemm_split(C.byref(d), K._ptr(ws), ws.numel(), K._stream()), "split"))
        fl = 2.0 * Cin * M * N * H * W
        byts = 4.0 * N * H * W * (Cin + 2 * M)
        dY = torch.randn((N, M, H, W), device="cuda", generator=gen) * 1e-3
        t_wf = timeit(lambda: K.conv1x1_wgrad(X, dY))
        t_ws = timeit(lambda: K.conv1x1_wgrad(X, dY, split=True))
        print("%-30s fp32 %.3f ms (%.0f TF/s, %.2f TB/s) | split %.3f ms (%.0f TF/s equiv) | split/fp32 %.2f || filter "
              "gradient fp32 %.3f ms | split %.3f ms | %.2f" % (
                  name, t_f, fl / t_f / 1e9, byts / t_f / 1e9, t_s, fl / t_s / 1e9, t_s / t_f, t_wf, t_ws, t_ws / t_wf),
              flush=True)


if __name__ == "__main__":
    main()
