"""Is a bad (item, wave half) = its own result + the PREVIOUS item's accumulators (stale accumulators)?  P3 only, no bias / ReLU."""
import sys
import torch
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
tiles_y, tiles_x, mblocks, G = 5, 7, 2, 256
def item_of(n, mb, ty, tx):
    t = (n * tiles_y + ty) * tiles_x + tx
    seq = (t // 8) * mblocks + mb
    return seq * 8 + (t % 8)
def tile_of(it):
    xcd, seq = it & 7, it >> 3
    mb = seq % mblocks
    t = (seq // mblocks) * 8 + xcd
    tx = t % tiles_x; t //= tiles_x
    ty = t % tiles_y; n = t // tiles_y
    return n, mb, ty, tx
bad = (got - ref).abs() > 1e-4 * ref.abs().max()
idx = bad.nonzero()
seen = set()
for n, m, y, x in idx.tolist():
    key = (n, m // 32, y // 16, x // 16, (y % 16) // 8)
    if key in seen: continue
    seen.add(key)
    if len(seen) > 12: break
    n_, g_, ty, tx, wp = key
    mb = g_ // 2
    it = item_of(n_, mb, ty, tx)
    sl = (slice(g_ * 32, g_ * 32 + 32), slice(ty * 16 + wp * 8, ty * 16 + wp * 8 + 8), slice(tx * 16, tx * 16 + 16))
    d = (got[n_][sl] - ref[n_][sl])
    line = "item %d (n %d grp %d ty %d tx %d wp %d): |diff| max %.3f mean %.3f ;" % (it, n_, g_, ty, tx, wp, d.abs().max(), d.abs().mean())
    for back in (1, 2):
        pit = it - back * G
        if pit < 0: continue
        pn, pmb, pty, ptx = tile_of(pit)
        g2 = pmb * 2 + (g_ % 2)
        for gg in (g2, pmb * 2, pmb * 2 + 1):
            sl2 = (slice(gg * 32, gg * 32 + 32), slice(pty * 16 + wp * 8, pty * 16 + wp * 8 + 8), slice(ptx * 16, ptx * 16 + 16))
            p = ref[pn][sl2]
            line += " vs prev%d grp%d: max|d - p| %.3f;" % (back, gg, (d - p).abs().max())
    print(line)
