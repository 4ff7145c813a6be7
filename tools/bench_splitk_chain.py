"""Round 5, review item 1: split K on the two GEMMs whose fp32 output row is read next by a LayerNorm launch (out-proj -> LN1+LN2,
FFN2 -> LN3), the S partial planes summed by that launch (fdm_gemm_args.ksplit / fdm_ln_args.x_planes).

Two chains at cfg2's shape (4 clips x 200 frames = 800 rows, d = 1024), each replayed as one hipGraph:
  A. 12 x {out-proj-shaped GEMM 64x64 (+bias, +fp32 residual) -> LayerNorm}                  (the review's probe; kill criterion:
     the pair not >= 1.0 us faster at S = 2)
  B. 8 decoder layers {QKV -> attention -> out-proj -> LN1+LN2 -> FFN1 -> FFN2 -> LN3} with 8 distinct weight sets
for S in {1, 2, 4} and the ring depths of the 64-column tiles.   python tools/bench_splitk_chain.py [bf16|f16x3] [rows]
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
code = {'bf16': BF16, 'f16x3': F16X3, 'f32': F32}[mode]
B, L, d, H, ffn = 4, int(sys.argv[2]) if len(sys.argv) > 2 else 200, 1024, 8, 2048
M, hd = B * L, 128
torch.manual_seed(0)
TN = {TILE_64x64: "64x64/4", TILE_64x64_S2: "64x64/2", TILE_32x64_S3: "32x64/3"}      # (the 3-stage 64x64 ring of the committed runs -- commit 7a4671f -- is retired)


def opnd(rows, cols):
    if code == F32: return torch.zeros(rows, cols, device=DEV)
    return ops.Split.empty(rows, cols, code, DEV) if ops.is_split(code) else torch.zeros(rows, cols, device=DEV, dtype=torch.bfloat16)


def weight(n, k):
    return ops.to_operand((torch.randn(n, k, device=DEV) / math.sqrt(k)), code)


def raw(t):
    return t.planes if isinstance(t, ops.Split) else t


NW = 12
Ws = [weight(d, d) for _ in range(NW)]
bias = torch.randn(d, device=DEV) * 0.1
g1 = torch.ones(d, device=DEV); b1 = torch.zeros(d, device=DEV)
h = torch.randn(M, d, device=DEV); ht = ops.to_operand(h, code)
h0 = h.clone(); ht0 = raw(ops.to_operand(h0, code)).clone()
x1 = torch.empty(8, M, d, device=DEV)
PS = M * d


def ks(S):
    return dict(ksplit=S, ksplit_stride=PS) if S > 1 else {}


def lp(S):
    return dict(x_planes=S, x_plane_stride=PS) if S > 1 else {}


def chain_a(S, tile):
    for i in range(NW):
        ops.gemm(ht, Ws[i], M, d, d, bias=bias, resid=h, out_f32=x1, tile=tile, **ks(S))
        ops.layernorm(x1, g1, b1, M, d, y_f32=h, y_t=(None if code == F32 else ht), dtype=code, **lp(S))


def reset():
    h.copy_(h0)
    if code != F32: raw(ht).copy_(ht0)


ref = None
print(f"== {mode}, {M} rows ==")
for S, tile in ((1, TILE_64x64), (2, TILE_64x64), (2, TILE_64x64_S2), (4, TILE_64x64), (4, TILE_64x64_S2), (2, TILE_32x64_S3), (4, TILE_32x64_S3)):
    reset(); chain_a(S, tile); torch.cuda.synchronize(); out = h.clone()
    reset(); chain_a(S, tile); torch.cuda.synchronize(); det = bool(torch.equal(out, h))
    if ref is None: ref = out
    us = timeit(lambda: chain_a(S, tile), n_rec=1, reps=50)
    print(f"A. 12 x (GEMM K=1024 + LayerNorm), S={S} tile {TN[tile]:8s}: {us / NW:7.2f} us per pair | max|diff| vs S=1 {float((out - ref).abs().max()):.2e} | deterministic {det}")

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
    kp, vp, _ = ops.kv_buffers(B, H, L, hd, torch.float32 if code == F32 else torch.bfloat16, DEV)
slopes = torch.tensor([2.0 ** -(i + 1) for i in range(H)], device=DEV)
C1 = torch.randn(M, d, device=DEV) * 0.1
split = ops.is_split(code)
t_qkv, t_ffn1 = TILE_80x128, (TILE_64x128 if split else TILE_64x64)


def layers(So, to, Sf, tf):
    for l in range(NL):
        kw = dict(out_f32=q) if code == F32 else dict(out_t=q, ldo_t=d)
        ops.gemm(ht, Wqkv[l], M, 3 * d, d, bias=bqkv, out_kp=kp, kp_col0=d, out_vp=vp, vp_col0=2 * d,
                 kv_L=L, kv_Lpad=Lpad, kv_hd=hd, tile=t_qkv, **kw)
        ops.attention(q, kp, vp, ctx, B=B, H=H, L=L, hd=hd, ldq=d, ldo=d, Lpad=Lpad, scale=1 / math.sqrt(hd), causal=True,
                      slopes=slopes, period=30)
        ops.gemm(ctx, Wo[l], M, d, d, bias=bias, resid=h, out_f32=x1, tile=to, **ks(So))
        ops.layernorm(x1, g1, b1, M, d, add_mat=C1, gamma2=g1, beta2=b1, y_f32=h2, y_t=(None if code == F32 else h2t), dtype=code, **lp(So))
        kw = dict(out_f32=u) if code == F32 else dict(out_t=u)
        ops.gemm(h2t if code != F32 else h2, W1[l], M, ffn, d, bias=b_ffn, act=ACT_RELU, tile=t_ffn1, **kw)
        ops.gemm(u, W2[l], M, d, ffn, bias=bias, resid=h2, out_f32=x1, tile=tf, **ks(Sf))
        ops.layernorm(x1, g1, b1, M, d, y_f32=h, y_t=(None if code == F32 else ht), dtype=code, **lp(Sf))


variants = [(1, TILE_64x64, 1, TILE_64x64),
            (2, TILE_64x64, 1, TILE_64x64), (2, TILE_64x64_S2, 1, TILE_64x64), (4, TILE_64x64_S2, 1, TILE_64x64),
            (1, TILE_64x64, 2, TILE_64x64), (1, TILE_64x64, 2, TILE_64x64_S2), (1, TILE_64x64, 4, TILE_64x64), (1, TILE_64x64, 4, TILE_64x64_S2),
            (2, TILE_64x64, 2, TILE_64x64), (2, TILE_64x64, 4, TILE_64x64), (2, TILE_64x64, 4, TILE_64x64_S2), (2, TILE_64x64_S2, 4, TILE_64x64_S2),
            (2, TILE_32x64_S3, 4, TILE_32x64_S3), (4, TILE_64x64_S2, 4, TILE_64x64_S2),
            (4, TILE_32x64_S3, 4, TILE_32x64_S3), (4, TILE_32x64_S3, 8, TILE_32x64_S3), (2, TILE_32x64_S3, 8, TILE_32x64_S3), (8, TILE_32x64_S3, 8, TILE_32x64_S3),
            (4, TILE_64x64, 8, TILE_64x64)]
ref = None
for So, to, Sf, tf in variants:
    reset(); layers(So, to, Sf, tf); torch.cuda.synchronize()
    out = h.clone()
    if ref is None: ref = out
    us = timeit(lambda: layers(So, to, Sf, tf), n_rec=1, reps=30)
    print(f"B. 8 layers, out-proj S={So} {TN[to]:8s} | FFN2 S={Sf} {TN[tf]:8s}: {us / NL:7.2f} us per layer | max|diff| vs S=1 {float((out - ref).abs().max()):.2e}"
          f" | finite {bool(torch.isfinite(out).all())}")
