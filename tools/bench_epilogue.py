"""Fixed cost of a GEMM launch (K = 64: one k-tile) by epilogue content and tile: where do the extra microseconds of the
large tiles go?"""
import sys, math, torch
sys.path.insert(0, 'face-diffusion-model_amd'); sys.path.insert(0, 'tools')
from fdm_amd import ops
from bench_ops import timeit
DEV = 'cuda:0'
dt = torch.bfloat16
for M in (800, 1992):
    N, K = 1024, 64
    A = torch.randn(M, K, device=DEV).to(dt); W = (torch.randn(N, K, device=DEV) / 8).to(dt)
    bias = torch.randn(N, device=DEV); res = torch.randn(M, N, device=DEV)
    o32 = torch.empty(M, N, device=DEV); ot = torch.empty(M, N, device=DEV, dtype=dt)
    for tile, name in ((1, "64x64"), (2, "128x64"), (3, "128x128"), (5, "256x128")):
        r = []
        r.append(timeit(lambda: ops.gemm(A, W, M, N, K, out_t=ot, tile=tile)))
        r.append(timeit(lambda: ops.gemm(A, W, M, N, K, out_f32=o32, tile=tile)))
        r.append(timeit(lambda: ops.gemm(A, W, M, N, K, bias=bias, out_f32=o32, tile=tile)))
        r.append(timeit(lambda: ops.gemm(A, W, M, N, K, bias=bias, resid=res, out_f32=o32, tile=tile)))
        r.append(timeit(lambda: ops.gemm(A, W, M, N, K, bias=bias, resid=res, out_f32=o32, out_t=ot, tile=tile)))
        print(f"M={M} {name:8s}: bf16 out {r[0]:5.2f} | f32 out {r[1]:5.2f} | +bias {r[2]:5.2f} | +resid {r[3]:5.2f} | +both outs {r[4]:5.2f} us")
