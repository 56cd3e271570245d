"""Isolated timing of the fused bias(+residual)+ReLU pass and the stem pool on the
bottleneck geometries at batch 16: achieved HBM GB/s.   python tools/tail_probe.py"""
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
    for (C, H, W) in [(256, 160, 224), (64, 160, 224), (512, 80, 112), (128, 80, 112), (1024, 40, 56),
                      (256, 40, 56), (2048, 20, 28), (512, 20, 28)]:
        z = torch.randn(N, C, H, W, device="cuda")
        r = torch.randn(N, C, H, W, device="cuda")
        b = torch.randn(C, device="cuda")
        nb = z.numel() * 4
        t1 = timeit(lambda: K.affine_channel_(z, b, residual=r, relu=True))
        t2 = timeit(lambda: K.affine_channel_(z, b, residual=None, relu=True))
        t3 = timeit(lambda: torch.relu_(z.add_(r)))
        print("%4d x %3dx%3d  bias+res+relu %.3f ms %5.0f GB/s | bias+relu %.3f ms %5.0f GB/s | torch add_+relu_ %.3f ms"
              % (C, H, W, t1, 3 * nb / t1 / 1e6, t2, 2 * nb / t2 / 1e6, t3), flush=True)
    z = torch.randn(N, 64, 320, 448, device="cuda")
    b = torch.randn(64, device="cuda")
    t = timeit(lambda: K.max_pool3x3s2_bias_relu(z, b))
    print("stem pool %.3f ms %5.0f GB/s" % (t, z.numel() * 5 / t / 1e6))
    dy = torch.randn(N, 256, 160, 224, device="cuda")
    y = torch.relu(torch.randn(N, 256, 160, 224, device="cuda"))
    t = timeit(lambda: K.relu_grad(y, dy))
    print("relu_grad 587 MB: %.3f ms %5.0f GB/s" % (t, 3 * dy.numel() * 4 / t / 1e6))


if __name__ == "__main__":
    main()
