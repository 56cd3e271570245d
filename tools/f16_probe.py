"""Timing of the fp16-storage subnet convolution on the bs-16 subnet geometries.
    python tools/f16_probe.py"""
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
    for (ci, co, H, W) in [(256, 256, 80, 112), (256, 256, 40, 56), (256, 256, 20, 28), (256, 720, 80, 112),
                           (256, 36, 80, 112), (720, 256, 80, 112)]:
        x = torch.randn(N, ci, H, W, device="cuda")
        w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
        b = torch.randn(co, device="cuda")
        xb = K.f16_pack_activations(x)
        w16, _ = K.f16_pack_filter(w, True, False)
        nchw = co % 8 != 0 or co == 720
        t = timeit(lambda: K.conv3x3_forward_f16(xb, w16, b, ci, co, relu=True, out_nchw_f32=nchw))
        fl = 2.0 * 9 * N * H * W * ci * co / 1e9
        print("%3d->%3d @%3dx%3d %6.1f GF  fp16 forward %.3f ms  %5.0f TF/s (%.1f %% of 2500)"
              % (ci, co, H, W, fl, t, fl / t, fl / t / 25.0), flush=True)
    for (ci, co, H, W) in [(256, 256, 80, 112), (256, 256, 40, 56), (256, 720, 80, 112), (256, 36, 80, 112)]:
        xb = K.f16_pack_activations(torch.randn(N, ci, H, W, device="cuda"))
        dyb = K.f16_pack_activations(torch.randn(N, co, H, W, device="cuda"))
        t = timeit(lambda: K.conv3x3_wgrad_f16([xb], [dyb], ci, co))
        fl = 2.0 * 9 * N * H * W * ci * co / 1e9
        print("%3d->%3d @%3dx%3d %6.1f GF  fp16 wgrad   %.3f ms  %5.0f TF/s (%.1f %% of 2500)"
              % (ci, co, H, W, fl, t, fl / t, fl / t / 25.0), flush=True)
    x = torch.randn(N, 256, 80, 112, device="cuda")
    t = timeit(lambda: K.f16_pack_activations(x))
    print("pack 147 MB fp32 -> blocked fp16: %.3f ms" % t)


if __name__ == "__main__":
    main()
