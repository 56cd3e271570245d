"""A/B of the persistent Winograd kernel's split tail (ssad_conv_wino_split_tail) on the backbone's 3x3 shapes at
bs 16, isolated: ms per launch with and without.    python tools/split_tail_ab.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
import ssad_amd  # noqa
from ssad_amd import kernels as K

SHAPES = [(128, 128, 80, 112), (256, 256, 40, 56), (512, 512, 20, 28), (256, 256, 80, 112), (256, 256, 20, 28),
          (256, 256, 10, 14), (256, 720, 40, 56), (256, 720, 80, 112), (2048, 256, 20, 28)]


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    N, L = 16, K.lib()
    for (ci, co, H, W) in SHAPES:
        x = torch.randn(N, ci, H, W, device="cuda")
        w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
        b = torch.randn(co, device="cuda")
        wf, _ = K.conv_wino_pack_filter(w, True, False)
        out = [torch.empty(N, co, H, W, device="cuda")]
        fl = 2.0 * 9 * N * H * W * ci * co / 1e9
        fn = lambda: K.conv3x3_forward([x], wf, b, co, relu=True, wino=True, out=out)
        res = {}
        for on in (0, 1, 0, 1):
            L.ssad_conv_wino_split_tail(on)
            res.setdefault(on, []).append(timeit(fn))
        arr = K._conv_levels([x], out, None)
        L.ssad_conv_wino_split_tail(1)
        nl = L.ssad_conv3x3_forward_wino_launches_for(arr, 1, co, ci, 0)
        t0, t1 = min(res[0]), min(res[1])
        print("%4d->%3d @%3dx%3d %6.1f GF | unsplit %.3f ms %5.1f TF/s exec | split %.3f ms %5.1f TF/s exec (%d launches) | %+.1f %%"
              % (ci, co, H, W, fl, t0, fl / t0 / 2.25, t1, fl / t1 / 2.25, nl, 100.0 * (t1 - t0) / t0), flush=True)


if __name__ == "__main__":
    main()
