"""Isolated timing of the 3x3 forward engines on the subnets' shapes (bs 16, 600 px pyramid): the split-operand engine
(conv3x3_split.hip: |max| + split + convolution launches, all inside the timed region) against Winograd F(2x4) /
F(2x2) fp32.  python tools/split_bench.py [--what towers|cls|res]"""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssad_amd  # noqa: E402,F401
from ssad_amd import kernels as K  # noqa: E402


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
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    args = ap.parse_args()
    K.lib()
    N = args.n
    gen = torch.Generator(device="cuda").manual_seed(1)
    shapes = [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)]
    cases = [("tower 256->256 x4 towers, 5 levels", 256, 256, shapes, 4, True),
             ("tower 256->256 one filter, 5 levels", 256, 256, shapes, 1, True),
             ("cls_pred 256->720, 5 levels", 256, 720, shapes, 1, False),
             ("dgrad 720->256, 5 levels", 720, 256, shapes, 1, False),
             ("res3 128->128 80x112", 128, 128, shapes[:1], 1, True),
             ("res4 256->256 40x56", 256, 256, shapes[1:2], 1, True),
             ("res5 512->512 20x28", 512, 512, shapes[2:3], 1, True)]
    for name, C, M, shp, reps, relu in cases:
        Xs = [torch.randn((N, C, h, w), device="cuda", generator=gen).clamp_(min=0) for h, w in shp] * reps
        Wt = torch.randn((M, C, 3, 3), device="cuda", generator=gen) * 0.02
        b = torch.randn(M, device="cuda", generator=gen)
        ps, p24 = K.conv_split_pack_filter(Wt), K.conv_wino24_pack_filter(Wt)
        p22, _ = K.conv_wino_pack_filter(Wt, True, False)
        outs = [torch.empty((N, M, x.shape[2], x.shape[3]), device="cuda") for x in Xs]
        L = K.lib()
        arr = K._conv_levels(Xs, outs, None)
        ws = torch.empty(L.ssad_conv3x3_split_workspace_bytes(arr, len(Xs), C), dtype=torch.uint8, device="cuda")
        flops = 2.0 * 9 * C * M * N * sum(h * w for h, w in shp) * reps
        t_s = timeit(lambda: K.conv3x3_forward_split(Xs, ps, b, M, relu=relu, out=outs, workspace=ws))
        t_24 = timeit(lambda: K.conv3x3_forward_wino24(Xs, p24, b, M, relu=relu, out=outs))
        t_22 = timeit(lambda: K.conv3x3_forward(Xs, p22, b, M, relu=relu, out=outs, wino=True))
        print("%-40s split %.3f ms (%.0f TF/s direct-equiv, %.2f of the fp16 peak on 3x the flops) | F(2x4) %.3f ms | "
              "F(2x2) %.3f ms | split/F(2x4) %.2f" % (name, t_s, flops / t_s / 1e9, 3 * flops / t_s / 1e9 / 2500.0, t_24,
                                                       t_22, t_s / t_24), flush=True)


if __name__ == "__main__":
    main()
