import numpy as np
rng=np.random.default_rng(0)
def mats(pts=(0,1,-1,2,-2)):
    # Lavin F(4x4,3x3) standard matrices
    BT=np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],np.float64)
    G=np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],np.float64)
    AT=np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],np.float64)
    return BT,G,AT
def mats22():
    BT=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],np.float64)
    G=np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],np.float64)
    AT=np.array([[1,1,1,0],[0,1,-1,-1]],np.float64)
    return BT,G,AT
def wino(X,W,BT,G,AT,m):
    # X [C,H,W] padded already by 1; W [M,C,3,3]; fp32 arithmetic everywhere
    f=np.float32
    BT,G,AT=BT.astype(f),G.astype(f),AT.astype(f)
    C,Hp,Wp=X.shape; M=W.shape[0]; a=m+2
    H,Wd=Hp-2,Wp-2
    U=np.einsum('ij,mcjk,lk->ilmc',G,W.astype(f),G).astype(f)   # [a,a,M,C]  (fp32 storage of the packed filter)
    Y=np.zeros((M,H,Wd),f)
    for ty in range(0,H,m):
        for tx in range(0,Wd,m):
            d=X[:,ty:ty+a,tx:tx+a].astype(f)
            V=np.einsum('ij,cjk,lk->ilc',BT,d,BT).astype(f)     # [a,a,C]
            # products accumulated over C in fp32 sequentially in chunks of 4 (MFMA k=4) -> emulate with float32 cumulative sum
            Mx=np.zeros((a,a,M),f)
            for c0 in range(0,C,4):
                Mx+= np.einsum('ijmc,ijc->ijm',U[:,:,:,c0:c0+4],V[:,:,c0:c0+4]).astype(f)
            y=np.einsum('ij,jkm,lk->mil',AT,Mx,AT).astype(f)
            Y[:,ty:ty+m,tx:tx+m]=y
    return Y
def direct64(X,W):
    C,Hp,Wp=X.shape; M=W.shape[0]; H,Wd=Hp-2,Wp-2
    Y=np.zeros((M,H,Wd))
    for i in range(3):
        for j in range(3):
            Y+=np.einsum('mc,chw->mhw',W[:,:,i,j].astype(np.float64),X[:,i:i+H,j:j+Wd].astype(np.float64))
    return Y
for name,C,M,H,Wd,ws,relu in (("tower 256->256",256,256,16,16,0.01,True),("dgrad-like randn",256,256,16,16,0.01,False),("res3 128->128",128,128,16,16,0.06,True),("cls_pred 256->720",256,64,16,16,0.01,True)):
    X=rng.standard_normal((C,H+2,Wd+2)).astype(np.float32)
    if relu: X=np.maximum(X,0)
    X[:,0,:]=0;X[:,-1,:]=0;X[:,:,0]=0;X[:,:,-1]=0
    W=(rng.standard_normal((M,C,3,3))*ws).astype(np.float32)
    ref=direct64(X,W)
    for tag,(BT,G,AT),m in (("F(2x2)",mats22(),2),("F(4x4)",mats(),4)):
        Y=wino(X,W,BT,G,AT,m).astype(np.float64)
        err=np.abs(Y-ref); mx=np.abs(ref).max()
        tol=1e-4*np.abs(ref)+1e-5*mx
        print(name,tag,"max abs err/max %.2e"%(err.max()/mx),"rms err/rms %.2e"%(np.sqrt((err**2).mean())/np.sqrt((ref**2).mean())),"outside tol: %d / %d"%((err>tol).sum(),err.size), "worst err/tol %.2f"%(err/tol).max())
