"""Pointwise-convolution probe: MIOpen's F.conv2d against torch.matmul (rocBLAS /
hipBLASLt strided-batched GEMM), untuned and with TunableOp's per-shape pick, on the
1x1 geometries of the ResNet bottlenecks at 640x896, batch 16.

    python tools/gemm1x1_probe.py [--tune] [--csv gpurun_out/tunableop.csv]
"""
import argparse
import time

import torch
import torch.nn.functional as F

SHAPES = [  # (Cin, Cout, H, W)
    (64, 64, 160, 224), (64, 256, 160, 224), (256, 64, 160, 224),
    (256, 128, 160, 224), (128, 512, 80, 112), (512, 128, 80, 112),
    (512, 256, 80, 112), (256, 1024, 40, 56), (1024, 256, 40, 56),
    (1024, 512, 40, 56), (512, 2048, 20, 28), (2048, 512, 20, 28),
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
    ap.add_argument("--tune", action="store_true")
    ap.add_argument("--csv", default="gpurun_out/tunableop.csv")
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    if a.tune:
        import torch.cuda.tunable as T
        T.enable(True)
        T.tuning_enable(True)
        T.set_filename(a.csv)
        T.set_max_tuning_duration(15)
        T.set_max_tuning_iterations(5)
    N = a.batch
    tot = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    for (ci, co, H, W) in SHAPES:
        x = torch.randn(N, ci, H, W, device="cuda")
        w = torch.randn(co, ci, 1, 1, device="cuda") * 0.05
        dy = torch.randn(N, co, H, W, device="cuda")
        w2 = w.view(co, ci)
        P = H * W
        f_mi = timeit(lambda: F.conv2d(x, w))
        f_mm = timeit(lambda: torch.matmul(w2, x.view(N, ci, P)))
        d_mi = timeit(lambda: torch.ops.aten.convolution_backward(
            dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]))
        d_mm = timeit(lambda: torch.matmul(w2.t(), dy.view(N, co, P)))
        g_bmm = timeit(lambda: torch.bmm(dy.view(N, co, P), x.view(N, ci, P).transpose(1, 2)).sum(0))
        # one GEMM over K = N*P by treating the batch as part of the reduction needs a
        # [co, N*P] view, which NCHW does not give; baddbmm-free alternative: einsum
        g_ein = timeit(lambda: torch.einsum("nmp,ncp->mc", dy.view(N, co, P), x.view(N, ci, P)))
        fl = 2.0 * N * P * ci * co / 1e9
        print("%4d->%4d @%3dx%3d  fwd miopen %.3f matmul %.3f | dgrad miopen %.3f matmul %.3f | "
              "wgrad bmm %.3f einsum %.3f ms   (%.1f GF; fwd %.0f / %.0f TF/s)" %
              (ci, co, H, W, f_mi, f_mm, d_mi, d_mm, g_bmm, g_ein, fl, fl / f_mi, fl / f_mm), flush=True)
        for i, v in enumerate((f_mi, f_mm, d_mi, d_mm, g_bmm, g_ein)):
            tot[i] += v
    print("sum: fwd miopen %.2f matmul %.2f | dgrad miopen %.2f matmul %.2f | wgrad bmm %.2f einsum %.2f ms" % tuple(tot))
    if a.tune:
        import torch.cuda.tunable as T
        T.write_file(a.csv) if hasattr(T, "write_file") else None
        print("results:", len(T.get_results()))


if __name__ == "__main__":
    main()
