import sys, numpy as np, torch
sys.path.insert(0, ".")
import ssad_amd
from ssad_amd import kernels as K
L = K.lib()
N, Cin, Cout, H, W = 1, 16, 128, 8, 16
X = torch.arange(N * Cin * H * W, device="cuda", dtype=torch.float32).reshape(N, Cin, H, W) + 1000.0
Wt = torch.zeros((Cout, Cin, 3, 3), device="cuda")
for m in range(Cout):
    Wt[m, m % Cin, 1, 1] = 1.0
wf, _ = K.conv_wino_pack_filter(Wt, True, False)
L.ssad_conv_wino_split_tail(0)
Yu = K.conv3x3_forward([X], wf, None, Cout, wino=True)[0].clone()
print("unsplit exact:", bool(torch.equal(Yu, X[:, [m % Cin for m in range(Cout)]])))
L.ssad_conv_wino_split_tail(1)
for rep in range(3):
    Ys = K.conv3x3_forward([X], wf, None, Cout, wino=True)[0].clone()
    bad = (Ys != Yu).nonzero().cpu().numpy()
    print("rep", rep, "bad", len(bad))
    for (n, m, y, x) in bad[:12]:
        got, want = Ys[n, m, y, x].item(), Yu[n, m, y, x].item()
        src = int(round(got - 1000.0))
        c, rem = divmod(src, H * W) if 0 <= src < Cin * H * W else (-1, 0)
        print("   m %3d (c %2d) y %d x %2d: want %.1f got %.3f  -> if a single input: c %d y %d x %d ; diff %.3f" % (m, m % Cin, y, x, want, got, c, rem // W, rem % W, got - want))
