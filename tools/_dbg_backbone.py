import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, numpy as np
import ssad_amd
from ssad_amd.backbone_pipeline import NativeResNetFPN
from ssad_amd.harness import full_model as fm
fm._HIP3X3 = fm._FUSE_TAIL = fm._GEMM_1X1 = fm._FUSED_PW = False
torch.manual_seed(11)
ref = fm.ResNetFPN("r50").cuda()
N, hw = 2, (256, 384)
nat = NativeResNetFPN("r50", N, hw, "cuda", train=True, src=ref, lr=0.01)
gen = torch.Generator(device="cuda").manual_seed(5)
images = torch.randn((N, 3) + hw, device="cuda", generator=gen)
d_fpn = None
res = {}
for mode in ("clean", "poison"):
    if mode == "poison":
        nat.poison()
    else:
        for t in nat._bufs: t.zero_()
    nat.pack()
    out = nat.forward(images)
    if d_fpn is None:
        d_fpn = [torch.randn(t.shape, device="cuda", generator=gen) for t in out]
    nat.backward(d_fpn)
    torch.cuda.synchronize()
    res[mode] = (nat.grads_flat.clone(), [t.clone() for t in out])
gc, gp = res["clean"][0], res["poison"][0]
print("fpn out nan:", [bool(torch.isnan(t).any()) for t in res["poison"][1]])
print("grads nan count", int(torch.isnan(gp).sum()), "of", gp.numel())
bad = []
for name, l in nat._layers.items():
    if l.train:
        for kind, g in (("w", l.gw), ("b", l.gb)):
            n = int(torch.isnan(g).sum())
            if n: bad.append((name, kind, n, g.numel()))
print("layers with NaN grads:", bad[:40])
ok = ~torch.isnan(gp)
print("max diff clean vs poison (non-nan):", float((gc[ok]-gp[ok]).abs().max()))
