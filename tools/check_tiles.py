"""Every GEMM tile / main-loop variant must give the same bits (same k order): compare each FDM_TILE_* against tile 1."""
import sys, math
sys.path.insert(0, 'face-diffusion-model_amd')
import torch
from fdm_amd import ops
from fdm_amd._lib import BF16, F32, ACT_RELU
DEV = 'cuda:0'
g = torch.Generator().manual_seed(0)
bad = 0
for dt in (BF16, F32):
    for (M, N, K) in ((800, 1024, 1024), (1992, 3072, 1024), (130, 1024, 2048), (2400, 512, 512), (77, 192, 256), (6400, 1024, 2048),
                      (300, 256, 64), (515, 384, 128), (1000, 128, 192)):
        A = ops.to_operand(torch.randn(M, K, generator=g).to(DEV), dt)
        W = ops.to_operand((torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV), dt)
        bias = torch.randn(N, generator=g).to(DEV)
        res = torch.randn(M, N, generator=g).to(DEV)
        ref = torch.zeros(M, N, device=DEV)
        ops.gemm(A, W, M, N, K, bias=bias, act=ACT_RELU, resid=res, out_f32=ref, tile=1)
        for tile in list(range(2, 13)) + [0x200 | t for t in (1, 2, 3, 8, 9, 11, 12)]:      # (0x200 = FDM_TILE_LOCKSTEP: no loader waves)
            out = torch.zeros(M, N, device=DEV)
            for rep in range(3):        # races would show as run-to-run differences
                ops.gemm(A, W, M, N, K, bias=bias, act=ACT_RELU, resid=res, out_f32=out, tile=tile)
                if not torch.equal(out, ref):
                    bad += 1
                    print(f"MISMATCH dtype {dt} shape {(M, N, K)} tile {tile} rep {rep}: max diff {float((out - ref).abs().max())}")
                    break
print("tile check:", "all identical" if not bad else f"{bad} mismatches")
