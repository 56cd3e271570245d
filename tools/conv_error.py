import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
import ssad_amd
from ssad_amd import kernels as K
import torch.nn.functional as F
torch.manual_seed(0)
for (N,C,M,H,W,ws) in [(2,256,256,40,56,0.01),(2,256,256,40,56,0.05),(2,256,720,20,28,0.01),(2,720,256,20,28,0.01)]:
    X = torch.randn(N,C,H,W,device="cuda"); Wt = torch.randn(M,C,3,3,device="cuda")*ws; b = torch.randn(M,device="cuda")*0.1
    ref = F.conv2d(X.double(), Wt.double(), b.double(), padding=1)
    pf,_ = K.conv_pack_filter(Wt, True, False); wf,_ = K.conv_wino_pack_filter(Wt, True, False)
    yd = K.conv3x3_forward([X], pf, b, M)[0].double(); yw = K.conv3x3_forward([X], wf, b, M, wino=True)[0].double()
    ym = F.conv2d(X, Wt, b, padding=1).double()
    for name, y in (("direct", yd), ("wino", yw), ("miopen", ym)):
        e = (y-ref)
        print(N,C,M,H,W, name, "relL2 %.2e  max|e|/max|ref| %.2e" % (float(e.norm()/ref.norm()), float(e.abs().max()/ref.abs().max())))
