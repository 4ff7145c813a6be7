"""Round 5 experiment: XCD-aware 2-D tile order (fdm_gemm_args.xcd_band) for the plain GEMM grids.
 A. isolated GEMMs (8 distinct weights, graph replay) per shape x tile x band height
 B. the 8-layer decoder chain (QKV -> attention -> out-proj -> LN1+LN2 -> FFN1 -> FFN2 -> LN3) at a row count, band per site
    python tools/bench_xcd_band.py [bf16|f16x3] [clips] [frames]"""
import math
import sys

import torch

sys.path.insert(0, 'face-diffusion-model_amd'); sys.path.insert(0, 'tools')
from fdm_amd import ops
from fdm_amd._lib import *  # noqa: F401,F403
from bench_ops import timeit

DEV = 'cuda:0'
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = int(sys.argv[3]) if len(sys.argv) > 3 else 498
code = {'bf16': BF16, 'f16x3': F16X3}[mode]
d, H, ffn = 1024, 8, 2048
M, hd = B * L, 128
torch.manual_seed(0)
split = ops.is_split(code)
TN = {1: "64x64", 2: "128x64", 3: "128x128", 7: "128x64/3", 11: "80x128", 12: "64x128", 5: "256x128"}


def opnd(rows, cols):
    return ops.Split.empty(rows, cols, code, DEV) if split else torch.zeros(rows, cols, device=DEV, dtype=torch.bfloat16)


def weight(n, k):
    return ops.to_operand((torch.randn(n, k, device=DEV) / math.sqrt(k)), code)


print(f"== {mode}, {M} rows ==")
for (n, k, tiles) in ((1024, 1024, (1, 2)), (1024, 2048, (1, 2)), (2048, 1024, (1, 2, 3, 12)), (3072, 1024, (2, 3, 7, 11)), (1024, 4096, (2,)), (4096, 1024, (3, 5))):
    A = ops.to_operand(torch.randn(M, k, device=DEV), code)
    Ws = [weight(n, k) for _ in range(8)]
    bias = torch.randn(n, device=DEV); res = torch.randn(M, n, device=DEV); o32 = torch.empty(M, n, device=DEV); ref = torch.empty(M, n, device=DEV)
    for tile in tiles:
        ops.gemm(A, Ws[0], M, n, k, bias=bias, resid=res, out_f32=ref, tile=tile); torch.cuda.synchronize()
        line = f"A. {M} x {n} x {k} tile {TN[tile]:8s}:"
        for hb in (0, 1, 2, 3, 4, 6, 8, 16):
            prog = ops.Program()
            with prog:
                for j in range(8):
                    ops.gemm(A, Ws[j], M, n, k, bias=bias, resid=res, out_f32=o32, tile=tile, xcd_band=hb)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                prog.instantiate(); prog.replay(3)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s); prog.replay(10); e1.record(s)
            s.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 80
            ops.gemm(A, Ws[0], M, n, k, bias=bias, resid=res, out_f32=o32, tile=tile, xcd_band=hb); torch.cuda.synchronize()
            line += f" h={hb}: {us:6.2f}{'' if torch.equal(o32, ref) else '!'}"
        print(line, flush=True)

NL = 8
Wqkv = [weight(3 * d, d) for _ in range(NL)]; Wo = [weight(d, d) for _ in range(NL)]
W1 = [weight(ffn, d) for _ in range(NL)]; W2 = [weight(d, ffn) for _ in range(NL)]
bias = torch.randn(d, device=DEV) * 0.1
bqkv = torch.randn(3 * d, device=DEV) * 0.1; b_ffn = torch.randn(ffn, device=DEV) * 0.1
g1 = torch.ones(d, device=DEV); b1 = torch.zeros(d, device=DEV)
h = torch.randn(M, d, device=DEV); ht = ops.to_operand(h, code)
x1 = torch.empty(M, d, device=DEV)
q = opnd(M, d); ctx = opnd(M, d); u = opnd(M, ffn); h2 = torch.empty(M, d, device=DEV); h2t = opnd(M, d)
Lpad = ops.kv_pad(L)
if split:
    kp = ops.Split(torch.zeros(2, B * H, Lpad * hd, device=DEV, dtype=torch.float16), code)
    vp = ops.Split(torch.zeros(2, B * H, Lpad * hd, device=DEV, dtype=torch.float16), code)
else:
    kp, vp, _ = ops.kv_buffers(B, H, L, hd, torch.bfloat16, DEV)
slopes = torch.tensor([2.0 ** -(i + 1) for i in range(H)], device=DEV)
C1 = torch.randn(M, d, device=DEV) * 0.1


def layers(hq, ho, h1, h2_):
    for l in range(NL):
        ops.gemm(ht, Wqkv[l], M, 3 * d, d, bias=bqkv, out_t=q, ldo_t=d, out_kp=kp, kp_col0=d, out_vp=vp, vp_col0=2 * d,
                 kv_L=L, kv_Lpad=Lpad, kv_hd=hd, xcd_band=hq)
        ops.attention(q, kp, vp, ctx, B=B, H=H, L=L, hd=hd, ldq=d, ldo=d, Lpad=Lpad, scale=1 / math.sqrt(hd), causal=True,
                      slopes=slopes, period=30)
        ops.gemm(ctx, Wo[l], M, d, d, bias=bias, resid=h, out_f32=x1, xcd_band=ho)
        ops.layernorm(x1, g1, b1, M, d, add_mat=C1, gamma2=g1, beta2=b1, y_f32=h2, y_t=h2t, dtype=code)
        ops.gemm(h2t, W1[l], M, ffn, d, bias=b_ffn, act=ACT_RELU, out_t=u, xcd_band=h1)
        ops.gemm(u, W2[l], M, d, ffn, bias=bias, resid=h2, out_f32=x1, xcd_band=h2_)
        ops.layernorm(x1, g1, b1, M, d, y_f32=h, y_t=ht, dtype=code)


for hs in ((0, 0, 0, 0), (2, 2, 2, 2), (4, 4, 4, 4), (8, 8, 8, 8), (4, 0, 0, 0), (0, 4, 0, 0), (0, 0, 4, 0), (0, 0, 0, 4), (3, 3, 3, 3), (6, 6, 6, 6)):
    us = timeit(lambda: layers(*hs), n_rec=1, reps=30)
    print(f"B. 8 layers (heuristic tiles), bands qkv/out/ffn1/ffn2 = {hs}: {us / NL:7.2f} us per layer", flush=True)
