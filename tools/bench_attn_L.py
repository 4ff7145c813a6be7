"""Attention launch time against the clip length (key tiles per wave): B = 4, H = 8, head_dim 128, bf16, causal ALiBi -- the step's launch at L = 200 gives each of a
workgroup's four waves up to two 32-key tiles (two dependent fragment round trips); L <= 128 gives one.  20 launches in one graph, HIP events."""
import math, sys, torch
sys.path.insert(0, 'face-diffusion-model_amd')
from fdm_amd import ops
sys.path.insert(0, 'tools')
from bench_ops import timeit
DEV = 'cuda:0'
B, H, hd = 4, 8, 128
d = H * hd
for L in (64, 96, 128, 129, 160, 200, 256, 257, 300, 384):
    M = B * L
    Lpad = ops.kv_pad(L)
    q = torch.randn(M, d, device=DEV).bfloat16()
    kp = torch.randn(B * H, Lpad * hd, device=DEV).bfloat16(); vp = torch.randn(B * H, Lpad * hd, device=DEV).bfloat16()
    o = torch.empty(M, d, device=DEV, dtype=torch.bfloat16); sl = torch.tensor([2.0 ** -(i + 1) for i in range(H)], device=DEV)
    us = timeit(lambda: ops.attention(q, kp, vp, o, B=B, H=H, L=L, hd=hd, ldq=d, ldo=d, Lpad=Lpad, scale=1 / math.sqrt(hd), causal=True, slopes=sl, period=30))
    nt = (L + 31) // 32
    print(f"attention L={L:4d}: {us:6.2f} us   key tiles {nt:2d} -> up to {(nt + 3) // 4} per wave, {B * H * ((L + 15) // 16)} workgroups")
