"""A few launches of the pointwise-convolution GEMM kernels at backbone shapes (bs 16), for
rocprofv3 --pmc passes (tools/profile_round.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssad_amd  # noqa: E402,F401
from ssad_amd import kernels as K  # noqa: E402

N = 16
for (ci, co, H, W) in [(1024, 256, 40, 56), (256, 1024, 40, 56), (512, 128, 80, 112), (2048, 512, 20, 28)]:
    x = torch.randn(N, ci, H, W, device="cuda")
    w = torch.randn(co, ci, 1, 1, device="cuda") * 0.05
    dy = torch.randn(N, co, H, W, device="cuda")
    wt = K.transpose_filter(w)
    y = torch.empty(N, co, H, W, device="cuda")
    dw = torch.empty(co, ci, device="cuda")
    for _ in range(5):
        K.conv1x1_forward(x, wt, co, out=y)
        K.conv1x1_wgrad(x, dy, out=dw)
    torch.cuda.synchronize()
