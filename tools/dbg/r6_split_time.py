"""conv3x3_split_kernel at config 3's shapes (bs 16), launches timed together and (SPLIT_PARTS=1) the |max| + split
passes alone: tower depth (4 filters x 5 levels), tower dgrad (2 x 5, masked), cls_pred, cls_pred dgrad, res4."""
import os, sys, time
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
        pf = K.conv_split_pack_filter(w)
        for h, wd in shapes:
            xs.append(torch.randn(N, ci, h, wd, device="cuda", generator=g)); packs.append(pf)
            outs.append(torch.empty(N, co, h, wd, device="cuda"))
            masks.append(torch.randn(N, co, h, wd, device="cuda", generator=g) if masked else None)
    L = K.lib()
    arr = K._conv_levels(xs, outs, masks if masked else None, packs, [None] * len(xs))
    ws = torch.empty(L.ssad_conv3x3_split_workspace_bytes(arr, len(xs), ci), dtype=torch.uint8, device="cuda")
    flags = K.CONV_MASK_AUX if masked else K.CONV_RELU
    fn = lambda: K._check(L.ssad_conv3x3_forward_split(arr, len(xs), K._ptr(packs[0]), None, co, ci, flags, K._ptr(ws),
                                                       ws.numel(), None, None, K._stream()), "split")
    ms = timeit(fn)
    fl = 2.0 * 9 * ci * co * N * sum(h * w for h, w in shapes) * nprob
    print("%-34s %8.3f ms   %6.1f TF/s direct-equivalent   3x flops / fp16 peak %.3f" % (name, ms, fl / ms / 1e9, 3 * fl / ms / 1e9 / 2500.0), flush=True)

case("tower depth (4 x 5 levels)", 4, 256, 256, lv)
if os.environ.get("SPLIT_ALL", "1") == "1":
    case("tower data gradient (2 x 5, mask)", 2, 256, 256, lv, masked=True)
    case("cls_pred 256->720 (5 levels)", 1, 256, 720, lv)
    case("res4 256->256 @40x56", 1, 256, 256, [(40, 56)])
    case("cls_pred data gradient 720->256 (5 levels, mask)", 1, 720, 256, lv, masked=True)
    case("512->256 (5 levels)", 1, 512, 256, lv)
