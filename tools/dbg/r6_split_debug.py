"""Where does conv3x3_split_kernel differ from the F(2x2) engine at full size?  (mismatch map per level / image / tile)"""
import sys
import torch
sys.path.insert(0, ".")
import ssad_amd
from ssad_amd import kernels as K
gen = torch.Generator(device="cuda").manual_seed(26)
N, C, M = 16, 256, 256
shapes = [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)]
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 5
shapes = shapes[:nl]
Xs = [torch.randn((N, C, h, w), device="cuda", generator=gen).clamp_(min=0) for h, w in shapes]
Wt = torch.randn((M, C, 3, 3), device="cuda", generator=gen) * 0.02
b = torch.randn(M, device="cuda", generator=gen)
ps = K.conv_split_pack_filter(Wt)
p22, _ = K.conv_wino_pack_filter(Wt, True, False)
Y22 = K.conv3x3_forward(Xs, p22, b, M, relu=True, wino=True)
for rep in range(3):
    Ys = K.conv3x3_forward_split(Xs, ps, b, M, relu=True)
    torch.cuda.synchronize()
    for l, (a, c) in enumerate(zip(Ys, Y22)):
        bad = (a - c).abs() > 1e-4 * c.abs().max()
        nb = int(bad.sum())
        if nb == 0:
            print("rep", rep, "level", l, "ok")
            continue
        idx = bad.nonzero()
        n_, m_, y_, x_ = idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]
        tiles = set(zip(n_.tolist(), (m_ // 128).tolist(), (y_ // 16).tolist(), (x_ // 16).tolist()))
        print("rep", rep, "level", l, "bad", nb, "of", bad.numel(), "tiles (n, mb, ty, tx):", len(tiles), sorted(tiles)[:12])
        t0 = sorted(tiles)[0]
        sub = bad[t0[0], t0[1] * 128:(t0[1] + 1) * 128, t0[2] * 16:(t0[2] + 1) * 16, t0[3] * 16:(t0[3] + 1) * 16]
        print("   first tile: bad per channel-32 group", [int(sub[i * 32:(i + 1) * 32].sum()) for i in range(4)],
              "per row", [int(sub[:, r].sum()) for r in range(sub.shape[1])])
