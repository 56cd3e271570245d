"""Fused pointwise conv + bias (+ shortcut) + ReLU against the library route (tuned GEMM + the
fused tail pass) on the early-stage bottleneck shapes, batch 16.   python tools/pw_probe.py"""
import sys
import time
import torch
sys.path.insert(0, ".")
import ssad_amd  # noqa
from ssad_amd import kernels as K
from tools.harness import full_model as fm


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
    fm.setup_tunableop()
    N = 16
    for (ci, co, H, W, res) in [(64, 256, 160, 224, True), (256, 64, 160, 224, False), (64, 64, 160, 224, False),
                                (128, 512, 80, 112, True), (512, 128, 80, 112, False),
                                (256, 1024, 40, 56, True), (1024, 256, 40, 56, False)]:
        x = torch.randn(N, ci, H, W, device="cuda")
        w = torch.randn(co, ci, 1, 1, device="cuda") * 0.05
        b = torch.randn(co, device="cuda")
        r = torch.randn(N, co, H, W, device="cuda") if res else None
        ref = torch.relu(torch.nn.functional.conv2d(x, w, b) + (r if res else 0))
        ok = co % 128 == 0
        if ok:
            y = K.conv1x1_bias_act(x, w, b, r, relu=True)
            err = float((y - ref).abs().max() / ref.abs().max())
            t_f = timeit(lambda: K.conv1x1_bias_act(x, w, b, r, relu=True))
        t_l = timeit(lambda: K.affine_channel_(fm._mm1x1(x, w), b, residual=r, relu=True))
        gb = (x.numel() + (2 if res else 1) * N * co * H * W) * 4 / 1e9
        if ok:
            print("%4d->%4d @%3dx%3d res=%d  fused %.3f ms (%.0f GB/s, err %.1e) | GEMM + tail %.3f ms"
                  % (ci, co, H, W, res, t_f, gb / t_f * 1e3, err, t_l), flush=True)
        else:
            print("%4d->%4d @%3dx%3d res=%d  (M %% 128 != 0) | GEMM + tail %.3f ms" % (ci, co, H, W, res, t_l))


if __name__ == "__main__":
    main()
