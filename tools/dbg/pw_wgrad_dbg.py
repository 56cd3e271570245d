import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import ssad_amd
from ssad_amd import kernels as K
DEV="cuda"
g=torch.Generator(device=DEV).manual_seed(4)
N,Cc,M,H,W=2,256,64,24,40
x=torch.randn((N,Cc,H,W),device=DEV,generator=g); dy=torch.randn((N,M,H,W),device=DEV,generator=g)
L=K.lib()
nb=L.ssad_conv1x1_wgrad_f16_workspace_bytes(N,Cc,H,W,M)
ws=torch.zeros(nb,dtype=torch.uint8,device=DEV)
dw=torch.empty((M,Cc),device=DEV); db=torch.empty((M,),device=DEV)
s=torch.cuda.current_stream().cuda_stream
xb=K.f16_pack_activations(x); dyb=K.f16_pack_activations(dy)
rc=L.ssad_conv1x1_wgrad_f16(xb.data_ptr(),dyb.data_ptr(),N,Cc,H,W,M,0,1.0,None,dw.data_ptr(),db.data_ptr(),ws.data_ptr(),nb,s)
torch.cuda.synchronize()
want=torch.einsum("nmhw,nchw->mc",dy.half().double(),x.half().double())
err=(dw.double()-want).abs()
bad=(err>1e-2*want.abs().max())
print("rc",rc,"nb",nb,"bad count",int(bad.sum()),"of",bad.numel())
idx=bad.nonzero()
print("bad rows",sorted(set(idx[:,0].tolist()))[:40])
print("bad cols",sorted(set(idx[:,1].tolist()))[:80])
print("db err",float((db.double()-dy.half().double().sum(dim=(0,2,3))).abs().max()))
# compare with the 3x3 kernel's centre tap
w3=torch.empty((M,Cc,3,3),device=DEV)
nb3=L.ssad_conv3x3_wgrad_f16_workspace_bytes(N,Cc,H,W,M)
ws3=torch.zeros(nb3,dtype=torch.uint8,device=DEV)
rc=L.ssad_conv3x3_wgrad_f16(xb.data_ptr(),dyb.data_ptr(),N,Cc,H,W,M,0,1.0,w3.data_ptr(),None,ws3.data_ptr(),nb3,s)
torch.cuda.synchronize()
e3=(w3[:,:,1,1].double()-want).abs().max()
print("3x3 centre tap err",float(e3))
