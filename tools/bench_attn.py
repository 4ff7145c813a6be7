import sys, math, torch
sys.path.insert(0, 'face-diffusion-model_amd')
sys.path.insert(0, 'tools')
from fdm_amd import ops
from bench_ops import timeit
DEV = 'cuda:0'
dt = torch.bfloat16
d, H, hd = 1024, 8, 128
for (B, L, causal) in ((4, 16, True), (4, 32, True), (4, 64, True), (4, 200, True), (4, 200, False), (50, 16, True), (4, 600, True)):
    M = B * L; Lpad = (L + 31) // 32 * 32
    qkv = torch.randn(M, d, device=DEV).to(dt)
    kp = torch.randn(B * H, Lpad * hd, device=DEV).to(dt); vp = torch.randn(B * H, Lpad * hd, device=DEV).to(dt)
    o = torch.empty(M, d, device=DEV, dtype=dt); sl = torch.tensor([2.0 ** -(i + 1) for i in range(H)], device=DEV)
    us = timeit(lambda: ops.attention(qkv, kp, vp, o, B=B, H=H, L=L, hd=hd, ldq=d, ldo=d, Lpad=Lpad,
                                      scale=1 / math.sqrt(hd), causal=causal, slopes=sl if causal else None, period=30))
    print(f"attention B={B} L={L} causal={causal}: {us:6.2f} us  blocks={((L+15)//16)*H*B}")
