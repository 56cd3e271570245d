"""Isolated timing of FPN's stride-2 3x3 layers (P6: 2048 -> 256 on 20x28, P7: 256 -> 256 on 10x14; bs 16) at their
own size (implicit GEMM with split-K / conv_strided.hip) against the stride-1 Winograd layer + subsampling.
    python tools/strided_probe.py"""
import sys
import time
import torch
sys.path.insert(0, ".")
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


def main():
    N = 16
    for (ci, co, H, W) in [(2048, 256, 20, 28), (256, 256, 10, 14), (2048, 256, 16, 24), (256, 256, 8, 12)]:
        x = torch.randn(N, ci, H, W, device="cuda")
        w = torch.randn(co, ci, 3, 3, device="cuda") * 0.02
        b = torch.randn(co, device="cuda")
        oh, ow = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        dy = torch.randn(N, co, oh, ow, device="cuda")
        fl = 2.0 * 9 * N * oh * ow * ci * co / 1e9
        y = torch.empty(N, co, oh, ow, device="cuda")
        t_f = timeit(lambda: K.conv_implicit_gemm(x, w, b, stride=2, pad=1, out=y, split_k=True))
        t_f1 = timeit(lambda: K.conv_implicit_gemm(x, w, b, stride=2, pad=1, out=y, split_k=False))
        t_w = timeit(lambda: K.conv_kxk_wgrad(x, dy, 3, 2, 1))
        t_d = timeit(lambda: K.conv_kxk_dgrad(w, dy, H, W, 2, 1))
        wf, wd = K.conv_wino_pack_filter(w, True, True)
        dyf = K.subsample_grad(dy, H, W, 2)
        o_f = timeit(lambda: K.subsample(K.conv3x3_forward([x], wf, b, co, wino=True)[0], 2))
        o_d = timeit(lambda: K.conv3x3_forward([K.subsample_grad(dy, H, W, 2)], wd, None, ci, wino=True))
        o_w = timeit(lambda: K.conv3x3_wgrad([x], [dyf], co))
        print("%4d->%3d @%2dx%2d %5.1f GF | fwd own %.3f ms (%3.0f TF/s; unsplit %.3f) wino+sub %.3f | dgrad own %.3f "
              "wino %.3f | wgrad own %.3f wino %.3f" % (ci, co, H, W, fl, t_f, fl / t_f, t_f1, o_f, t_d, o_d, t_w, o_w),
              flush=True)


if __name__ == "__main__":
    main()
