"""[needs the library of commit 2451829: the fused layer tail was measured 9-20 % slower on every shape and removed again -- profiles/README.md,
round 4, profiles/r4_fused_tail/]
Per-phase clock stamps of the fused layer tail (fdm_tail_args.stamps): 8 layers' tails back to back as one hipGraph (distinct
weights), stamps of the LAST replay, averaged over the 256 workgroups; beside it the five operators as separate launches.
    python tools/tail_probe.py [bf16|f16x3] [rows]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "face-diffusion-model_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fdm_amd import ops  # noqa: E402
from fdm_amd._lib import *  # noqa: F401,F403,E402
from bench_ops import timeit  # noqa: E402

DEV = "cuda:0"
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 800
code = {"bf16": BF16, "f16x3": F16X3}[mode]
d, ffn, NL = 1024, 2048, 8
torch.manual_seed(0)


def opnd(r, c):
    return ops.Split.empty(r, c, code, DEV) if ops.is_split(code) else torch.zeros(r, c, device=DEV, dtype=torch.bfloat16)


def weight(n, k):
    return ops.to_operand(torch.randn(n, k, device=DEV) / math.sqrt(k), code)


Wo = [weight(d, d) for _ in range(NL)]; W1 = [weight(ffn, d) for _ in range(NL)]; W2 = [weight(d, ffn) for _ in range(NL)]
bias = torch.randn(d, device=DEV) * 0.1; b1 = torch.randn(ffn, device=DEV) * 0.1
g = torch.ones(d, device=DEV); be = torch.zeros(d, device=DEV)
ctx = ops.to_operand(torch.randn(M, d, device=DEV), code)
h = torch.randn(M, d, device=DEV); ht = opnd(M, d); x1 = torch.empty(M, d, device=DEV)
h2 = torch.empty(M, d, device=DEV); h2t = opnd(M, d); u = opnd(M, ffn); C1 = torch.randn(M, d, device=DEV) * 0.1
sync = torch.zeros(16 * 32, device=DEV, dtype=torch.int32); err = torch.zeros(32, device=DEV, dtype=torch.int32)
stamps = torch.zeros(256 * 12, device=DEV, dtype=torch.int64)
h0 = h.clone()


def args(l):
    return (ops.gemm_args(ctx, Wo[l], M, d, d, bias=bias, resid=h, out_f32=x1),
            ops.ln_args(x1, g, be, M, d, add_mat=C1, y_f32=h2, y_t=h2t, dtype=code, gamma2=g, beta2=be),
            ops.gemm_args(h2t, W1[l], M, ffn, d, bias=b1, act=ACT_RELU, out_t=u),
            ops.gemm_args(u, W2[l], M, d, ffn, bias=bias, resid=h2, out_f32=x1),
            ops.ln_args(x1, g, be, M, d, y_f32=h, y_t=ht, dtype=code))


def fused(st=None):
    for l in range(NL):
        a = args(l)
        ops.layer_tail(a[0], a[1], a[2], a[3], a[4], M, sync, err, st)


def separate():
    for l in range(NL):
        ops.gemm(ctx, Wo[l], M, d, d, bias=bias, resid=h, out_f32=x1, tile=TILE_64x64)
        ops.layernorm(x1, g, be, M, d, add_mat=C1, gamma2=g, beta2=be, y_f32=h2, y_t=h2t, dtype=code)
        ops.gemm(h2t, W1[l], M, ffn, d, bias=b1, act=ACT_RELU, out_t=u)
        ops.gemm(u, W2[l], M, d, ffn, bias=bias, resid=h2, out_f32=x1, tile=TILE_64x64)
        ops.layernorm(x1, g, be, M, d, y_f32=h, y_t=ht, dtype=code)


h.copy_(h0); separate(); torch.cuda.synchronize(); ref = h.clone()
h.copy_(h0); fused(); torch.cuda.synchronize()
print(f"{mode} rows {M}: fused == separate bit for bit: {bool(torch.equal(ref, h))}; spin time-outs {int(err[0])}")
ts = timeit(separate, n_rec=1, reps=30) / NL
tf = timeit(lambda: fused(None), n_rec=1, reps=30) / NL
print(f"five launches {ts:.2f} us per layer tail | one fused launch {tf:.2f} us")
timeit(lambda: fused(stamps), n_rec=1, reps=5)
torch.cuda.synchronize()
s = stamps.view(256, 12).cpu().double()
names = ["entry->ticket", "out-proj", "barrier 1", "LN1+LN2", "barrier 2", "FFN1", "barrier 3", "FFN2", "barrier 4", "LN3 (+stores acked)"]
dd = (s[:, 1:11] - s[:, 0:10]) / 100.0
print("phase (us): mean over 256 workgroups | max")
for i, n in enumerate(names):
    print(f"  {n:22s} {dd[:, i].mean():6.2f} | {dd[:, i].max():6.2f}")
print(f"  total entry -> end     {((s[:, 10] - s[:, 0]) / 100.0).mean():6.2f} | span first entry -> last end {(s[:, 10].max() - s[:, 0].min()) / 100.0:6.2f}")
