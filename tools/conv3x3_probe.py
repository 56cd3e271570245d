"""Per-shape timing of this repo's 3x3 kernels (Winograd forward, Winograd weight
gradient) against MIOpen through torch, on the backbone's and the subnets' geometries
at batch 16.    python tools/conv3x3_probe.py"""
import sys
import time
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import ssad_amd  # noqa
from ssad_amd import kernels as K

SHAPES = [(64, 64, 160, 224), (128, 128, 80, 112), (256, 256, 40, 56), (512, 512, 20, 28),
          (256, 256, 80, 112), (256, 256, 20, 28), (256, 256, 10, 14), (256, 720, 80, 112), (256, 36, 80, 112)]


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
    N = 16
    for (ci, co, H, W) in SHAPES:
        x = torch.randn(N, ci, H, W, device="cuda")
        w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
        b = torch.randn(co, device="cuda")
        dy = torch.randn(N, co, H, W, device="cuda")
        wf, wd = K.conv_wino_pack_filter(w, True, True)
        fl = 2.0 * 9 * N * H * W * ci * co / 1e9
        t_f = timeit(lambda: K.conv3x3_forward([x], wf, b, co, relu=True, wino=True))
        t_d = timeit(lambda: K.conv3x3_forward([dy], wd, None, ci, wino=True))
        t_w = timeit(lambda: K.conv3x3_wgrad([x], [dy], co))
        m_f = timeit(lambda: F.conv2d(x, w, b, 1, 1))
        m_b = timeit(lambda: torch.ops.aten.convolution_backward(
            dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, True, False]))
        xb = K.f16_pack_activations(x)
        w16, _ = K.f16_pack_filter(w, True, False)
        t_h = timeit(lambda: K.conv3x3_forward_f16(xb, w16, b, ci, co, relu=True, out_nchw_f32=co % 8 != 0))
        print("      fp16 storage forward %.3f ms %5.0f TF/s" % (t_h, fl / t_h), flush=True)
        print("%3d->%3d @%3dx%3d %6.1f GF | fwd %.3f ms %5.0f TF/s (miopen %.3f) | dgrad %.3f %5.0f | wgrad %.3f %5.0f | "
              "miopen dgrad+wgrad %.3f" % (ci, co, H, W, fl, t_f, fl / t_f, m_f, t_d, fl / t_d, t_w, fl / t_w, m_b),
              flush=True)


if __name__ == "__main__":
    main()
