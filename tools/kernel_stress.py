"""Every hand-synchronised kernel family under contention (round 4; round 6: + the split-operand engines).  Each case runs repeatedly on the current stream
while two other streams keep the chip busy with GEMM-shaped, Winograd and streaming work; every output is compared bit
for bit with the output of a quiet run of the same call (all kernels are deterministic, so a difference is a
synchronisation bug that only shows when memory is late -- how the in-flight-ring corruption of wino_conv_z_kernel
was reproduced).      python tools/kernel_stress.py [iterations]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssad_amd  # noqa
from ssad_amd import kernels as K

def run(iters=150, verbose=True):
    prev = K.lib().ssad_conv_wino_split_tail(2)     # round 5: every partial round of the Winograd grid is split (default: tiny launches only)
    try:
        return _run(iters, verbose)
    finally:
        K.lib().ssad_conv_wino_split_tail(prev)


def _run(iters, verbose):
    N = 16
    g = torch.Generator(device="cuda").manual_seed(3)
    R = lambda *s: torch.randn(s, device="cuda", generator=g)


    def wino_case(Cin, Cout, H, W, mask=False, relu=True, bias=True):
        x = R(N, Cin, H, W)
        pf, _ = K.conv_wino_pack_filter(R(Cout, Cin, 3, 3) / (3 * Cin ** 0.5), True, False)
        b = R(Cout) if bias else None
        m = [R(N, Cout, H, W)] if mask else None
        out = [torch.empty(N, Cout, H, W, device="cuda")]
        return lambda: K.conv3x3_forward([x], pf, b, Cout, relu=relu, mask_by=m, wino=True, out=out)[0]


    def wino_levels_case(Cin, Cout, shapes):
        xs = [R(N, Cin, h, w) for h, w in shapes]
        pf, _ = K.conv_wino_pack_filter(R(Cout, Cin, 3, 3) / (3 * Cin ** 0.5), True, False)
        b = R(Cout)
        outs = [torch.empty(N, Cout, h, w, device="cuda") for h, w in shapes]
        return lambda: torch.cat([t.reshape(-1) for t in K.conv3x3_forward(xs, pf, b, Cout, relu=True, wino=True, out=outs)])


    def w24_case(Cin, Cout, shapes, mask=False, relu=True, bias=True, nfilters=1):
        """wino24_conv_kernel (round 5: the default forward / data-gradient engine of >= 128-wide layers): `nfilters`
        problems per level list in one launch, like a tower depth."""
        xs, packs, outs, masks, biases = [], [], [], [], []
        for _ in range(nfilters):
            pf = K.conv_wino24_pack_filter(R(Cout, Cin, 3, 3) / (3 * Cin ** 0.5))
            b = R(Cout) if bias else None
            for h, w in shapes:
                xs.append(R(N, Cin, h, w)); packs.append(pf); biases.append(b)
                outs.append(torch.empty(N, Cout, h, w, device="cuda"))
                masks.append(R(N, Cout, h, w) if mask else None)
        arr = K._conv_levels(xs, outs, masks if mask else None, packs, biases)
        flags = (K.CONV_MASK_AUX if mask else 0) | (K.CONV_RELU if relu else 0)
        L = K.lib()
        def run():
            K._check(L.ssad_conv3x3_forward_wino24(arr, len(xs), K._ptr(packs[0]), K._ptr(biases[0]), Cout, Cin, flags,
                                                   K._stream()), "conv3x3_forward_wino24")
            return torch.cat([t.reshape(-1) for t in outs])
        return run


    def split_case(Cin, Cout, shapes, mask=False, relu=True, bias=True, nfilters=1):
        """conv3x3_split_kernel (round 6: persistent, one wave per SIMD, hand-counted waits over a nine-tap filter ring and
        three LDS stages; several work items per workgroup at these sizes): `nfilters` problems per level list."""
        xs, packs, outs, masks, biases = [], [], [], [], []
        for _ in range(nfilters):
            pf = K.conv_split_pack_filter(R(Cout, Cin, 3, 3) / (3 * Cin ** 0.5))
            b = R(Cout) if bias else None
            for h, w in shapes:
                xs.append(R(N, Cin, h, w)); packs.append(pf); biases.append(b)
                outs.append(torch.empty(N, Cout, h, w, device="cuda"))
                masks.append(R(N, Cout, h, w) if mask else None)
        arr = K._conv_levels(xs, outs, masks if mask else None, packs, biases)
        flags = (K.CONV_MASK_AUX if mask else 0) | (K.CONV_RELU if relu else 0)
        L = K.lib()
        ws = torch.empty(L.ssad_conv3x3_split_workspace_bytes(arr, len(xs), Cin), dtype=torch.uint8, device="cuda")
        def run():
            K._check(L.ssad_conv3x3_forward_split(arr, len(xs), K._ptr(packs[0]), K._ptr(biases[0]), Cout, Cin, flags,
                                                  K._ptr(ws), ws.numel(), None, None, K._stream()), "conv3x3_forward_split")
            return torch.cat([t.reshape(-1) for t in outs])
        return run


    def gsplit_case(Cin, Cout, H, W, res=True):
        x, wt = R(N, Cin, H, W), K.transpose_filter(R(Cout, Cin, 1, 1) / Cin ** 0.5)
        b, r = R(Cout), (R(N, Cout, H, W) if res else None)
        out = torch.empty(N, Cout, H, W, device="cuda")
        return lambda: K.conv1x1_forward(x, wt, Cout, b, r, relu=True, out=out, split=True).reshape(-1)


    def wgrad_case(Cin, Cout, shapes, split=False):
        """split: wsplit_kernel (round 6: 8 waves, two LDS stages, chunk c + 2 fetched while chunk c multiplies)"""
        xs = [R(N, Cin, h, w) for h, w in shapes]
        dys = [R(N, Cout, h, w) for h, w in shapes]
        def run():
            r = K.conv3x3_wgrad(xs, dys, Cout, split=split)
            return torch.cat([t.reshape(-1) for t in (r if isinstance(r, (tuple, list)) else [r])])
        return run


    def pw_case(Cin, Cout, H, W, res=True):
        x, r = R(N, Cin, H, W), (R(N, Cout, H, W) if res else None)
        wt = K.transpose_filter(R(Cout, Cin, 1, 1) * 0.05)
        b = R(Cout)
        y = torch.empty(N, Cout, H, W, device="cuda")
        return lambda: K.conv1x1_forward(x, wt, Cout, bias=b, residual=r, relu=True, out=y)


    def pw_wgrad_case(Cin, Cout, H, W, split=False):
        x, dy = R(N, Cin, H, W), R(N, Cout, H, W)
        out = torch.empty(Cout, Cin, device="cuda")
        return lambda: K.conv1x1_wgrad(x, dy, out=out, split=split)


    def f16_case(Cin, Cout, H, W, mask=False):
        xb = K.f16_pack_activations(R(N, Cin, H, W))
        pf, _ = K.f16_pack_filter(R(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)) if hasattr(K, "f16_pack_filter") else (None, None)
        b = R(Cout)
        m = K.f16_pack_activations(R(N, Cout, H, W)) if mask else None
        return lambda: K.conv3x3_forward_f16(xb, pf, b, Cin, Cout, relu=not mask, mask_by=m)


    def f16_wgrad_case(Cin, Cout, H, W):
        xb, dyb = K.f16_pack_activations(R(N, Cin, H, W)), K.f16_pack_activations(R(N, Cout, H, W))
        def run():
            r = K.conv3x3_wgrad_f16([xb], [dyb], Cin, Cout)
            return torch.cat([t.reshape(-1) for t in r])
        return run


    CASES = [
        ("wino 64->64 160x224 (NHALF)", wino_case(64, 64, 160, 224)),
        ("wino 256->256 80x112 masked dgrad form", wino_case(256, 256, 80, 112, mask=True, relu=False, bias=False)),
        ("wino 256->256 five levels (pairs + patches)", wino_levels_case(256, 256, [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)])),
        ("wino 256->720 40x56 (pairs)", wino_case(256, 720, 40, 56, relu=False)),
        ("wino 256->36 80x112 (NHALF, 3 of 4 slices)", wino_case(256, 36, 80, 112, relu=False)),
        ("wino 512->512 20x28", wino_case(512, 512, 20, 28)),
        ("wino 256->256 40x56 (split tail: 48 items x 4 units)", wino_case(256, 256, 40, 56)),
        ("wino 256->256 40x56 masked (split tail)", wino_case(256, 256, 40, 56, mask=True, relu=False, bias=False)),
        ("wino 256->256 10x14 (no full round: 64 x 4 units)", wino_case(256, 256, 10, 14)),
        ("wino24 256->256 tower depth (2 filters x five levels)", w24_case(256, 256, [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)], nfilters=2)),
        ("wino24 256->256 data-gradient form, masked, five levels", w24_case(256, 256, [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)], mask=True, relu=False, bias=False)),
        ("wino24 256->720 40x56 + 5x7 (pairs)", w24_case(256, 720, [(40, 56), (5, 7)], relu=False)),
        ("wino24 720->256 80x112 masked (cls_pred data gradient)", w24_case(720, 256, [(80, 112)], mask=True, relu=False, bias=False)),
        ("wino24 512->512 20x28", w24_case(512, 512, [(20, 28)])),
        ("split 256->256 tower depth (2 filters x five levels)", split_case(256, 256, [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)], nfilters=2)),
        ("split 256->256 data-gradient form, masked, five levels", split_case(256, 256, [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)], mask=True, relu=False, bias=False)),
        ("split 256->720 five levels (cls_pred)", split_case(256, 720, [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)], relu=False)),
        ("split 720->256 80x112 masked (cls_pred data gradient)", split_case(720, 256, [(80, 112)], mask=True, relu=False, bias=False)),
        ("split gemm 1024->256 40x56 + shortcut", gsplit_case(1024, 256, 40, 56)),
        ("split gemm 2048->512 20x28", gsplit_case(2048, 512, 20, 28, res=False)),
        ("split gemm 256->1024 40x56 + shortcut (4 channel blocks per pixel tile)", gsplit_case(256, 1024, 40, 56)),
        ("split filter gradient 256x256 five levels", wgrad_case(256, 256, [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)], split=True)),
        ("split filter gradient 256x720 five levels (cls_pred)", wgrad_case(256, 720, [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)], split=True)),
        ("split filter gradient 512x512 20x28", wgrad_case(512, 512, [(20, 28)], split=True)),
        ("split pointwise filter gradient 1024x256 40x56", pw_wgrad_case(1024, 256, 40, 56, split=True)),
        ("split pointwise filter gradient 512x2048 20x28", pw_wgrad_case(512, 2048, 20, 28, split=True)),
        ("wino filter gradient 256x256 five levels", wgrad_case(256, 256, [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)])),
        ("wino filter gradient 128x128 80x112", wgrad_case(128, 128, [(80, 112)])),
        ("gemm nn 256->1024 40x56 + shortcut", pw_case(256, 1024, 40, 56)),
        ("gemm nn 64->256 160x224 + shortcut", pw_case(64, 256, 160, 224)),
        ("gemm nn 2048->512 20x28", pw_case(2048, 512, 20, 28, res=False)),
        ("gemm nt 1024x256 40x56", pw_wgrad_case(1024, 256, 40, 56)),
        ("gemm nt 64x256 160x224", pw_wgrad_case(64, 256, 160, 224)),
        ("fp16 3x3 256->256 64x96", f16_case(256, 256, 64, 96)),
        ("fp16 3x3 256->256 64x96 masked", f16_case(256, 256, 64, 96, mask=True)),
        ("fp16 filter gradient 256x256 64x96", f16_wgrad_case(256, 256, 64, 96)),
    ]

    # background load
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    bg_pw = pw_case(256, 1024, 80, 112)
    bg_w = wino_case(64, 64, 160, 224)
    bg_w2 = wino_case(256, 256, 40, 56)
    bg_w24 = w24_case(256, 256, [(40, 56)])
    bg_sp = split_case(256, 256, [(40, 56)])
    big = R(64, 1024, 1024)
    torch.cuda.synchronize()
    bad_total, skipped = 0, 0
    for name, fn in CASES:
        try:
            ref = fn().clone()
            torch.cuda.synchronize()
            assert torch.equal(fn(), ref), "quiet run not reproducible"
        except Exception as e:            # an API this build lacks: say so, go on
            print("%-50s skipped: %r" % (name, e), flush=True)
            skipped += 1
            continue
        bad, first = 0, None
        t0 = time.time()
        for it in range(iters):
            with torch.cuda.stream(s1):
                bg_pw(); bg_w(); bg_w2(); bg_w24(); bg_sp()
            with torch.cuda.stream(s2):
                big.mul_(1.0000001)
            out = fn()
            if not torch.equal(out, ref):
                bad += 1
                if first is None:
                    d = (out != ref)
                    first = (it, int(d.sum()), float((out - ref).abs().max()))
        torch.cuda.synchronize()
        bad_total += bad
        print("%-50s %4d / %d iterations differ (%.1f s)%s" % (name, bad, iters, time.time() - t0,
                                                              "" if not bad else "   first: %s" % (first,)), flush=True)
    if verbose:
        print("TOTAL differing iterations:", bad_total)
    return bad_total, len(CASES) - skipped


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 150)
