import sys, numpy as np, torch
sys.path.insert(0, ".")
import ssad_amd
from ssad_amd import kernels as K
L = K.lib()
def run(N, Cin, Cout, H, W):
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn((N, Cin, H, W), device="cuda", generator=g)
    Wt = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) * 0.05
    wf, _ = K.conv_wino_pack_filter(Wt, True, False)
    L.ssad_conv_wino_split_tail(0)
    Yu = K.conv3x3_forward([X], wf, None, Cout, wino=True)[0].clone()
    L.ssad_conv_wino_split_tail(1)
    outs = [K.conv3x3_forward([X], wf, None, Cout, wino=True)[0].clone() for _ in range(4)]
    d = (outs[0] - Yu).abs()
    bad = (d > 1e-3).nonzero().cpu().numpy()
    print("N%d %d->%d %dx%d: max diff %.3e, bad %d of %d, equal-to-first %s" % (N, Cin, Cout, H, W, d.max().item(), bad.shape[0], d.numel(),
          [bool(torch.equal(outs[0], o)) for o in outs[1:]]), flush=True)
    if bad.shape[0]:
        print("   m:", np.unique(bad[:, 1])[:40], " y:", np.unique(bad[:, 2]), " x:", np.unique(bad[:, 3]))
for cin in (16, 32, 48, 64, 128, 256):
    run(1, cin, 128, 8, 16)
run(1, 64, 16, 8, 16)
run(1, 64, 128, 8, 8)
