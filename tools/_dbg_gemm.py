import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import ssad_amd
from ssad_amd import kernels as K
torch.manual_seed(0)
for (N, Cin, M, H, W) in [(1, 16, 128, 8, 16), (2, 64, 64, 12, 16), (1, 32, 128, 16, 32)]:
    x = torch.randn(N, Cin, H, W, device="cuda")
    w = torch.randn(M, Cin, 1, 1, device="cuda")
    wt = K.transpose_filter(w)
    print("wt ok", torch.equal(wt[:, :M], w.view(M, Cin).t()), wt.shape)
    y = torch.full((N, M, H, W), 7.0, device="cuda")
    try:
        K.conv1x1_forward(x, wt, M, out=y)
    except Exception as e:
        print("ERR", e); continue
    torch.cuda.synchronize()
    ref = torch.einsum("mk,nkhw->nmhw", w.view(M, Cin), x)
    print((N, Cin, M, H, W), "max err", float((y - ref).abs().max()), "y sample", y.flatten()[:6].tolist(), "ref", ref.flatten()[:6].tolist())
    print("  count sevens", int((y == 7.0).sum()), "of", y.numel(), " zeros", int((y == 0).sum()))
