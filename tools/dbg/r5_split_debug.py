import sys, numpy as np, torch
sys.path.insert(0, ".")
import ssad_amd
from ssad_amd import kernels as K
L = K.lib()
def run(N, Cin, Cout, H, W):
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn((N, Cin, H, W), device="cuda", generator=g)
    Wt = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) * 0.05
    b = torch.randn(Cout, device="cuda", generator=g)
    wf, _ = K.conv_wino_pack_filter(Wt, True, False)
    L.ssad_conv_wino_split_tail(0)
    Yu = K.conv3x3_forward([X], wf, b, Cout, wino=True)[0].clone()
    L.ssad_conv_wino_split_tail(1)
    arr = K._conv_levels([X], [Yu], None)
    nl = L.ssad_conv3x3_forward_wino_launches_for(arr, 1, Cout, Cin, 0)
    outs = []
    for rep in range(3):
        Ys = K.conv3x3_forward([X], wf, b, Cout, wino=True)[0].clone()
        outs.append(Ys)
    d = (outs[0] - Yu).abs()
    bad = (d > 1e-3).nonzero()
    print("N%d %d->%d %dx%d launches %d: max diff %.3e, bad %d, reproducible %s" % (N, Cin, Cout, H, W, nl, d.max().item(), bad.shape[0],
          [bool(torch.equal(outs[0], o)) for o in outs[1:]]))
    if bad.shape[0]:
        b_ = bad.cpu().numpy()
        print("  bad n", np.unique(b_[:, 0]), "m range", b_[:, 1].min(), b_[:, 1].max(), "nunique m", len(np.unique(b_[:,1])), "y", np.unique(b_[:, 2]), "x", np.unique(b_[:, 3]))
for shp in [(1, 256, 256, 10, 14), (16, 256, 256, 10, 14), (1, 256, 256, 8, 16), (2, 256, 256, 16, 16), (16, 256, 256, 40, 56), (16, 512, 512, 20, 28), (1, 64, 128, 8, 16), (1, 128, 128, 10, 14)]:
    run(*shp)
