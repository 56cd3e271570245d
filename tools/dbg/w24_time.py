"""Times of wino24_conv_kernel launches at config 3's shapes (bs 16): one tower depth (4 filters x 5 levels), the
tower data gradient (2 filters, masked), cls_pred, res4 / res5 / res3 backbone layers; with the executed-flop
fraction of the fp32 MFMA peak (direct-form / 3)."""
import sys, time
import torch
sys.path.insert(0, ".")
import ssad_amd
from ssad_amd import kernels as K

def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

g = torch.Generator(device="cuda").manual_seed(3)
N = 16
lv = [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)]
def case(name, nprob, ci, co, shapes, masked=False):
    xs, packs, outs, masks = [], [], [], []
    for _ in range(nprob):
        w = torch.randn(co, ci, 3, 3, device="cuda", generator=g) * 0.02
        pf = K.conv_wino24_pack_filter(w)
        for h, wd in shapes:
            xs.append(torch.randn(N, ci, h, wd, device="cuda", generator=g)); packs.append(pf)
            outs.append(torch.empty(N, co, h, wd, device="cuda"))
            masks.append(torch.randn(N, co, h, wd, device="cuda", generator=g) if masked else None)
    L = K.lib()
    arr = K._conv_levels(xs, outs, masks if masked else None, packs, [None] * len(xs))
    flags = K.CONV_MASK_AUX if masked else K.CONV_RELU
    fn = lambda: K._check(L.ssad_conv3x3_forward_wino24(arr, len(xs), K._ptr(packs[0]), None, co, ci, flags, K._stream()), "w24")
    ms = timeit(fn)
    fl = 2.0 * 9 * ci * co * N * sum(h * w for h, w in shapes) * nprob
    print("%-34s %8.3f ms   %6.1f TF/s direct-equivalent   executed frac %.3f" % (name, ms, fl / ms / 1e9, fl / 3 / ms / 1e9 / 157.3), flush=True)

case("tower depth (4 x 5 levels)", 4, 256, 256, lv)
case("tower data gradient (2 x 5, mask)", 2, 256, 256, lv, masked=True)
case("cls_pred 256->720 (5 levels)", 1, 256, 720, lv)
case("cls_pred dgrad 720->256", 1, 720, 256, lv, masked=True)
case("res4 256->256 @40x56", 1, 256, 256, [(40, 56)])
case("res5 512->512 @20x28", 1, 512, 512, [(20, 28)])
case("res3 128->128 @80x112", 1, 128, 128, [(80, 112)])
