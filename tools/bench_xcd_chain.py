"""Round 4, review item 1: does a step-shaped operator chain run faster when every kernel consumes the rows ITS OWN XCD produced?

Two chains at cfg2's shape (4 clips x 200 frames = 800 rows, d = 1024), each replayed as one hipGraph, with the plain grids
and with the XCD-affine row map (fdm_xcd_map: every kernel's workgroup w works on row block w % 8):
  A. 12 x {out-proj-shaped GEMM 64x64 (+bias, +fp32 residual, fp32 + operand outputs) -> LayerNorm}   (the review's probe)
  B. 8 decoder layers {QKV -> attention -> out-proj -> LN1+LN2 -> FFN1 -> FFN2 -> LN3} with 8 distinct weight sets
Outputs of the two forms are compared bit for bit.   python tools/bench_xcd_chain.py [bf16|f16x3]
Needs the library of commit e27fd28 (fdm_xcd_map in the operator argument structs): the maps were measured and removed again
(profiles/README.md, round 4; results in profiles/r4_xcd_affinity/).
"""
import math
import sys

import torch

sys.path.insert(0, 'face-diffusion-model_amd'); sys.path.insert(0, 'tools')
from fdm_amd import ops
from fdm_amd._lib import *  # noqa: F401,F403
from bench_ops import timeit

DEV = 'cuda:0'
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
only = int(sys.argv[2]) if len(sys.argv) > 2 else -1      # profiling: run variant `only` of chain B alone (rocprofv3 --kernel-trace --stats)
code = {'bf16': BF16, 'f16x3': F16X3}[mode]
B, L, d, H, ffn = 4, 200, 1024, 8, 2048
M, hd = B * L, 128
torch.manual_seed(0)


def opnd(rows, cols):
    return ops.Split.empty(rows, cols, code, DEV) if ops.is_split(code) else torch.zeros(rows, cols, device=DEV, dtype=torch.bfloat16)


def weight(n, k):
    return ops.to_operand((torch.randn(n, k, device=DEV) / math.sqrt(k)), code)


xmap = ops.xcd_rows(B, L)
print("rows per XCD block:", [xmap.row0[i + 1] - xmap.row0[i] for i in range(8)])

# ---- A: GEMM -> LN chain --------------------------------------------------------------------------------------------------
NW = 12
Ws = [weight(d, d) for _ in range(NW)]
bias = torch.randn(d, device=DEV) * 0.1
g1 = torch.ones(d, device=DEV); b1 = torch.zeros(d, device=DEV)
h = torch.randn(M, d, device=DEV); ht = ops.to_operand(h, code)
h0 = h.clone(); ht0 = ops.to_operand(h0, code)
x1 = torch.empty(M, d, device=DEV)


def chain_a(xcd, tile=TILE_64x64):
    for i in range(NW):
        ops.gemm(ht, Ws[i], M, d, d, bias=bias, resid=h, out_f32=x1, tile=tile, xcd=xcd)
        ops.layernorm(x1, g1, b1, M, d, y_f32=h, y_t=ht, dtype=code, xcd=xcd)


def reset():
    h.copy_(h0)
    (ht.planes if ops.is_split(code) else ht).copy_(ht0.planes if ops.is_split(code) else ht0)


res = {}
for name, xcd in (("plain", None), ("xcd-affine", xmap)) if only < 0 else ():
    reset(); chain_a(xcd); torch.cuda.synchronize(); res[name] = h.clone()
    us = timeit(lambda: chain_a(xcd), n_rec=1, reps=50)
    print(f"A. {mode} 12 x (GEMM 64x64 + LayerNorm), {name:10s}: {us / NW:7.2f} us per pair")
if only < 0: print("A. bit-identical:", bool(torch.equal(res["plain"], res["xcd-affine"])))
# the GEMM alone in the chain: the same chain with the LayerNorm replaced by nothing is not a dependent chain of fresh operands,
# so time GEMM and LN shares by leaving one out of the replay with the other still producing its operand
for name, xcd in (("plain", None), ("xcd-affine", xmap)) if only < 0 else ():
    def only_ln():
        for i in range(NW): ops.layernorm(x1, g1, b1, M, d, y_f32=h, y_t=ht, dtype=code, xcd=xcd)
    print(f"A. {mode} LayerNorm alone x 12, {name:10s}: {timeit(only_ln, n_rec=1, reps=50) / NW:7.2f} us each")

# ---- B: decoder layers ----------------------------------------------------------------------------------------------------
NL = 8
Wqkv = [weight(3 * d, d) for _ in range(NL)]; Wo = [weight(d, d) for _ in range(NL)]
W1 = [weight(ffn, d) for _ in range(NL)]; W2 = [weight(d, ffn) for _ in range(NL)]
bqkv = torch.randn(3 * d, device=DEV) * 0.1; b_ffn = torch.randn(ffn, device=DEV) * 0.1
q = opnd(M, d); ctx = opnd(M, d); u = opnd(M, ffn); h2 = torch.empty(M, d, device=DEV); h2t = opnd(M, d)
Lpad = ops.kv_pad(L)
if ops.is_split(code):
    kp = ops.Split(torch.zeros(2, B * H, Lpad * hd, device=DEV, dtype=torch.float16), code)
    vp = ops.Split(torch.zeros(2, B * H, Lpad * hd, device=DEV, dtype=torch.float16), code)
else:
    kp, vp, _ = ops.kv_buffers(B, H, L, hd, torch.bfloat16, DEV)
slopes = torch.tensor([2.0 ** -(i + 1) for i in range(H)], device=DEV)
C1 = torch.randn(M, d, device=DEV) * 0.1


def layers(xcd, t_qkv, t_ffn1, t_sq=TILE_64x64):
    for l in range(NL):
        ops.gemm(ht, Wqkv[l], M, 3 * d, d, bias=bqkv, out_t=q, ldo_t=d, out_kp=kp, kp_col0=d, out_vp=vp, vp_col0=2 * d,
                 kv_L=L, kv_Lpad=Lpad, kv_hd=hd, tile=t_qkv, xcd=xcd)
        ops.attention(q, kp, vp, ctx, B=B, H=H, L=L, hd=hd, ldq=d, ldo=d, Lpad=Lpad, scale=1 / math.sqrt(hd), causal=True,
                      slopes=slopes, period=30, xcd=xcd)
        ops.gemm(ctx, Wo[l], M, d, d, bias=bias, resid=h, out_f32=x1, tile=t_sq, xcd=xcd)
        ops.layernorm(x1, g1, b1, M, d, add_mat=C1, gamma2=g1, beta2=b1, y_f32=h2, y_t=h2t, dtype=code, xcd=xcd)
        ops.gemm(h2t, W1[l], M, ffn, d, bias=b_ffn, act=ACT_RELU, out_t=u, tile=t_ffn1, xcd=xcd)
        ops.gemm(u, W2[l], M, d, ffn, bias=bias, resid=h2, out_f32=x1, tile=t_sq, xcd=xcd)
        ops.layernorm(x1, g1, b1, M, d, y_f32=h, y_t=ht, dtype=code, xcd=xcd)


split = ops.is_split(code)
variants = [("plain grids, the step's tiles (80x128, 64x64, 64x64)", None, TILE_80x128, TILE_64x128 if split else TILE_64x64),
            ("xcd-affine (112x128, 64x64, 64x128)", xmap, TILE_112x128, TILE_64x128),
            ("xcd-affine (64x128, 64x64, 64x128)", xmap, TILE_64x128, TILE_64x128),
            ("xcd-affine (112x128, 64x64, 64x64)", xmap, TILE_112x128, TILE_64x64),
            ("plain grids, the affine tiles (112x128, 64x64, 64x128)", None, TILE_112x128, TILE_64x128)]
ref = None
for name, xcd, tq, tf in (variants if only < 0 else variants[only:only + 1]):
    reset(); layers(xcd, tq, tf); torch.cuda.synchronize()
    out = h.clone()
    if ref is None: ref = out
    us = timeit(lambda: layers(xcd, tq, tf), n_rec=1, reps=30)
    print(f"B. {mode} 8 layers, {name}: {us / NL:7.2f} us per layer (7 launches) | bit-identical to the first: {bool(torch.equal(out, ref))}"
          f" | finite: {bool(torch.isfinite(out).all())}")
