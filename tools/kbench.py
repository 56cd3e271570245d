#!/usr/bin/env python3
"""Per-kernel timings at BASELINE sizes (bs=16/GPU, 600 px) on one MI355X.
Development aid; bench.py is the contract benchmark."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssad_amd  # noqa
from ssad_amd import kernels as K, synth


def timeit(fn, iters=20, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=16)
    ap.add_argument("--what", default="loss,conv")
    a = ap.parse_args()
    N, A, C = a.bs, 9, 80
    shapes = synth.LEVEL_SHAPES_600
    dev = "cuda"
    if "loss" in a.what:
        lv = []
        for h, w in shapes:
            x = torch.randn((N, A * C, h, w), device=dev) * 2 - 4
            q = torch.sigmoid(torch.randn((N, A * C, h, w), device=dev) * 2 - 4).clamp_(1e-6, 1 - 1e-6)
            g = torch.where(torch.rand((N, A, h, w), device=dev) < 0.05, -1, 0).to(torch.int32)
            lv.append((x, q, g))
        E = sum(x.numel() for x, _, _ in lv)
        norm = K.pow_sum([q for _, q, _ in lv], 1.8).reshape(1)
        kw = dict(gamma=2.0, alpha=0.5, beta=0.0, num_classes=C, ignored_label=-1, scale=1.0)
        t = timeit(lambda: K.pow_sum([q for _, q, _ in lv], 1.8))
        print("powsum       %8.3f ms  %7.1f GB/s" % (t, E * 4 / t / 1e6))
        t = timeit(lambda: K.distill_loss_forward(lv, norm, **kw))
        print("distill fwd  %8.3f ms  %7.1f GB/s (alg 8.05 B/elem)" % (t, E * 8.05 / t / 1e6))
        outs = [torch.empty_like(x) for x, _, _ in lv]
        one = torch.ones(5, device=dev)
        t = timeit(lambda: K.distill_loss_backward(lv, norm, one, out=outs, **kw))
        print("distill bwd  %8.3f ms  %7.1f GB/s (alg 12.05 B/elem)" % (t, E * 12.05 / t / 1e6))
        t = timeit(lambda: K.sigmoid(lv[0][0], out=outs[0]))
        print("sigmoid P3   %8.3f ms  %7.1f GB/s" % (t, lv[0][0].numel() * 8 / t / 1e6))
        del lv, outs
    if "wino" in a.what:
        for (M, Cin, name) in ((256, 256, "tower 256->256"), (720, 256, "cls_pred 256->720")):
            Xs = [torch.randn((N, Cin, h, w), device=dev) for h, w in shapes]
            Wt = torch.randn((M, Cin, 3, 3), device=dev) * 0.01
            b = torch.zeros(M, device=dev)
            px = sum(N * h * w for h, w in shapes)
            fl = 2.0 * 9 * M * Cin * px
            Ys = [torch.empty((N, M, h, w), device=dev) for h, w in shapes]
            wf, wd = K.conv_wino_pack_filter(Wt)
            t = timeit(lambda: K.conv3x3_forward(Xs, wf, b, M, relu=True, out=Ys, wino=True))
            print("%-18s WINO fwd all-lvl %8.3f ms  %6.1f TF/s (direct-equivalent)" % (name, t, fl / t / 1e9))
            t = timeit(lambda: K.conv3x3_forward(Xs[:1], wf, b, M, relu=True, out=Ys[:1], wino=True))
            print("%-18s WINO fwd P3 only %8.3f ms  %6.1f TF/s" % (name, t, 2.0 * 9 * M * Cin * N * 8960 / t / 1e9))
            del Xs, Ys
    if "conv" in a.what:
        for (M, Cin, name) in ((256, 256, "tower 256->256"), (720, 256, "cls_pred 256->720"),
                               (36, 256, "bbox_pred 256->36")):
            Xs = [torch.randn((N, Cin, h, w), device=dev) for h, w in shapes]
            dYs = [torch.randn((N, M, h, w), device=dev) for h, w in shapes]
            Wt = torch.randn((M, Cin, 3, 3), device=dev) * 0.01
            b = torch.zeros(M, device=dev)
            pf, pd = K.conv_pack_filter(Wt)
            px = sum(N * h * w for h, w in shapes)
            fl = 2.0 * 9 * M * Cin * px
            Ys = [torch.empty((N, M, h, w), device=dev) for h, w in shapes]
            dXs = [torch.empty((N, Cin, h, w), device=dev) for h, w in shapes]
            t = timeit(lambda: K.conv3x3_forward(Xs, pf, b, M, relu=True, out=Ys))
            print("%-18s fwd   all-levels %8.3f ms  %6.1f TF/s" % (name, t, fl / t / 1e9))
            t = timeit(lambda: K.conv3x3_forward(Xs[:1], pf, b, M, relu=True, out=Ys[:1]))
            print("%-18s fwd   P3 only    %8.3f ms  %6.1f TF/s" % (name, t, 2.0 * 9 * M * Cin * N * 8960 / t / 1e9))
            if M >= 32:
                wf, wd = K.conv_wino_pack_filter(Wt)
                t = timeit(lambda: K.conv3x3_forward(Xs, wf, b, M, relu=True, out=Ys, wino=True))
                print("%-18s WINO fwd all-lvl %8.3f ms  %6.1f TF/s (direct-equivalent)" % (name, t, fl / t / 1e9))
                t = timeit(lambda: K.conv3x3_forward(Xs[:1], wf, b, M, relu=True, out=Ys[:1], wino=True))
                print("%-18s WINO fwd P3 only %8.3f ms  %6.1f TF/s" % (name, t, 2.0 * 9 * M * Cin * N * 8960 / t / 1e9))
                t = timeit(lambda: K.conv3x3_forward(dYs, wd, None, Cin, out=dXs, wino=True))
                print("%-18s WINO dgrad all   %8.3f ms  %6.1f TF/s" % (name, t, fl / t / 1e9))
            t = timeit(lambda: K.conv3x3_forward(dYs, pd, None, Cin, out=dXs))
            print("%-18s dgrad all-levels %8.3f ms  %6.1f TF/s" % (name, t, fl / t / 1e9))
            dW = torch.empty_like(Wt)
            db = torch.empty(M, device=dev)
            t = timeit(lambda: K.conv3x3_wgrad(Xs, dYs, M, dW=dW, db=db))
            print("%-18s wgrad all-levels %8.3f ms  %6.1f TF/s (incl. reduce + dbias)" % (name, t, fl / t / 1e9))
            t = timeit(lambda: K.conv3x3_wgrad(Xs, dYs, M, dW=dW, want_db=False))
            print("%-18s wgrad no-dbias   %8.3f ms  %6.1f TF/s" % (name, t, fl / t / 1e9))
            t = timeit(lambda: K.conv_pack_filter(Wt))
            print("%-18s pack             %8.3f ms" % (name, t))
            del Xs, dYs, Ys, dXs
    if "f16" in a.what:
        for (M, Cin, name) in ((256, 256, "tower 256->256"), (720, 256, "cls_pred 256->720")):
            Xb = [K.f16_pack_activations(torch.randn((N, Cin, h, w), device=dev)) for h, w in shapes]
            dYb = [K.f16_pack_activations(torch.randn((N, M, h, w), device=dev)) for h, w in shapes]
            Wt = torch.randn((M, Cin, 3, 3), device=dev) * 0.01
            b = torch.zeros(M, device=dev)
            wf, wd = K.f16_pack_filter(Wt, True, True)
            px = sum(N * h * w for h, w in shapes)
            fl = 2.0 * 9 * M * Cin * px
            nchw = M == 720
            Ys = [torch.empty((N, M, h, w), device=dev) if nchw else
                  torch.empty((N, M // 8, h, w, 8), device=dev, dtype=torch.float16) for h, w in shapes]
            dXs = [torch.empty_like(x) for x in Xb]
            t = timeit(lambda: K.conv3x3_forward_f16_levels(Xb, wf, b, Cin, M, Ys, relu=not nchw,
                                                            out_nchw_f32=nchw))
            print("%-18s F16 fwd all-lvl  %8.3f ms  %6.1f TF/s" % (name, t, fl / t / 1e9))
            t = timeit(lambda: K.conv3x3_forward_f16_levels(dYb, wd, None, M, Cin, dXs))
            print("%-18s F16 dgrad all    %8.3f ms  %6.1f TF/s" % (name, t, fl / t / 1e9))
            t = timeit(lambda: K.conv3x3_wgrad_f16(Xb, dYb, Cin, M))
            print("%-18s F16 wgrad all    %8.3f ms  %6.1f TF/s (incl. reduce + dbias)" % (name, t, fl / t / 1e9))
            del Xb, dYb, Ys, dXs


if __name__ == "__main__":
    main()
