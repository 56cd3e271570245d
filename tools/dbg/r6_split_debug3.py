"""Which (chunk, tap) step of a bad (item, wave half) is wrong?  d = got - ref projected on each step's contribution."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import ssad_amd
from ssad_amd import kernels as K
gen = torch.Generator(device="cuda").manual_seed(26)
N, C, M = 16, 256, 256
H, W = 80, 112
X = torch.randn((N, C, H, W), device="cuda", generator=gen)
Wt = torch.randn((M, C, 3, 3), device="cuda", generator=gen) * 0.02
ps = K.conv_split_pack_filter(Wt)
p22, _ = K.conv_wino_pack_filter(Wt, True, False)
ref = K.conv3x3_forward([X], p22, None, M, wino=True)[0]
got = K.conv3x3_forward_split([X], ps, None, M)[0]
torch.cuda.synchronize()
bad = (got - ref).abs() > 1e-4 * ref.abs().max()
idx = bad.nonzero()
seen = set()
for n, m, y, x in idx.tolist():
    key = (n, m // 32, y // 16, x // 16, (y % 16) // 8)
    if key in seen: continue
    seen.add(key)
    if len(seen) > 8: break
    n_, g_, ty, tx, wp = key
    ys, xs = ty * 16 + wp * 8, tx * 16
    d = (got - ref)[n_, g_ * 32:g_ * 32 + 32, ys:ys + 8, xs:xs + 16].double()
    best = []
    Xp = F.pad(X[n_:n_ + 1].double(), (1, 1, 1, 1))
    for c in range(16):
        for t in range(9):
            dy, dx = t // 3, t % 3
            xs_ = Xp[0, c * 16:c * 16 + 16, ys + dy:ys + dy + 8, xs + dx:xs + dx + 16]          # [16][8][16]
            w = Wt[g_ * 32:g_ * 32 + 32, c * 16:c * 16 + 16, dy, dx].double()                       # [32][16]
            contrib = torch.einsum("mc,cyx->myx", w, xs_)
            coef = float((d * contrib).sum() / (contrib * contrib).sum())
            resid = float((d - coef * contrib).norm() / d.norm())
            best.append((resid, c, t, coef))
    best.sort()
    print(key, "|d| %.3f" % float(d.norm()), "best fits (resid, chunk, tap, coef):", [(round(r, 3), c, t, round(k, 3)) for r, c, t, k in best[:3]])
