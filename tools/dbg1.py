import sys, math, torch
sys.path.insert(0, 'face-diffusion-model_amd')
from fdm_amd import ops
from fdm_amd._lib import *
DEV='cuda:0'
torch.manual_seed(0)
for dtype in (F32, BF16):
    td = ops.tdtype(dtype)
    for (M,N,K) in ((64,64,32 if dtype==F32 else 64),(64,64,256),(7,256,256),(128,128,1024)):
        A = torch.randn(M,K).to(td); W = torch.randn(N,K).to(td)
        o = torch.zeros(M,N,device=DEV)
        ops.gemm(A.to(DEV), W.to(DEV), M,N,K, out_f32=o)
        torch.cuda.synchronize()
        ref = A.float()@W.float().t()
        err = (o.cpu()-ref).abs()
        print(dtype,M,N,K,'maxerr',float(err.max()),'ref max',float(ref.abs().max()))
        if err.max()>1e-2 and M<=64 and K<=64:
            # which k contribute? use one-hot probing
            for kk in range(K):
                A1 = torch.zeros(M,K); A1[:,kk]=1; W1=torch.zeros(N,K); W1[:,kk]=1
                o1 = torch.zeros(M,N,device=DEV)
                ops.gemm(A1.to(td).to(DEV), W1.to(td).to(DEV), M,N,K,out_f32=o1)
                print('k',kk,'count', float(o1[0,0]), float(o1.min()), float(o1.max()))
