"""HuBERT's positional conv (grouped Conv1d, k = 128, 16 groups; models/hubert.py:110-137) as a GEMM with overlapping A rows: one
launch per clip (round 4) against ONE launch for all clips with the groups pinned to XCDs (fdm_gemm_args.batch2, round 5), per tile.
    python tools/bench_posconv.py [bf16|f16x3] [clips] [frames] [D]"""
import math
import sys

import torch

sys.path.insert(0, 'face-diffusion-model_amd'); sys.path.insert(0, 'tools')
from fdm_amd import ops
from fdm_amd._lib import *  # noqa: F401,F403
from bench_ops import timeit

DEV = 'cuda:0'
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
T = int(sys.argv[3]) if len(sys.argv) > 3 else 498
D = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
code = {'bf16': BF16, 'f16x3': F16X3, 'f32': F32}[mode]
G, KW = 16, 128
dg = D // G
torch.manual_seed(0)
xg = ops.to_operand(torch.randn(G * C * (T + KW), dg, device=DEV), code)
wt = ops.to_operand(torch.randn(G * dg, KW * dg, device=DEV) / math.sqrt(KW * dg), code)
bias = torch.randn(D, device=DEV); resid = torch.randn(C * T, D, device=DEV)
kw = dict(lda=dg, batch=G, a_bs=C * (T + KW) * dg, w_bs=dg * KW * dg, bias=bias, bias_bs=dg, act=ACT_GELU_ERF, ldr=D, ldo_f32=D, out_bs=dg)
out = torch.zeros(C * T, D, device=DEV); ref = torch.zeros(C * T, D, device=DEV)


def per_clip(tile, dst=out):
    for c in range(C):
        ops.gemm(xg[c * (T + KW):], wt, T, dg, KW * dg, resid=resid[c * T:], out_f32=dst[c * T:], tile=tile, **kw)


def one(tile, dst=out):
    ops.gemm(xg, wt, T, dg, KW * dg, resid=resid, out_f32=dst, batch2=C, a_bs2=(T + KW) * dg, out_bs2=T * D, tile=tile, **kw)


per_clip(0, ref); torch.cuda.synchronize()
fl = 2.0 * C * T * D * KW * dg
print(f"== positional conv {mode}: {C} clips x {T} frames, D = {D} ({G} groups of {dg}), {fl / 1e9:.1f} GFLOP ==")
for name, tile in (("64x64", TILE_64x64), ("64x64/2", TILE_64x64_S2), ("128x64", TILE_128x64)):      # (profiles/r5_posconv/ also lists the 3-stage rings of commit 56559b2, since retired)
    a = timeit(lambda: per_clip(tile), n_rec=1, reps=20)
    out.zero_(); one(tile); torch.cuda.synchronize()
    b = timeit(lambda: one(tile), n_rec=1, reps=20)
    print(f"tile {name:9s}: one launch per clip {a:7.1f} us | one launch, groups pinned to XCDs {b:7.1f} us ({fl / b / 1e6:5.0f} TFLOP/s) | bit-identical {bool(torch.equal(out, ref))}")
