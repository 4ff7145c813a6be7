"""Do kernels cost more when they alternate (as in the step) than when the same kernel repeats (as in bench_ops)?"""
import sys, math, torch
sys.path.insert(0, 'face-diffusion-model_amd'); sys.path.insert(0, 'tools')
from fdm_amd import ops
from bench_ops import timeit
DEV = 'cuda:0'
dt = torch.bfloat16
M, d = 800, 1024
A = torch.randn(M, d, device=DEV).to(dt)
Ws = [(torch.randn(d, d, device=DEV) / 32).to(dt) for _ in range(10)]
W3 = (torch.randn(3 * d, d, device=DEV) / 32).to(dt)
o32 = torch.empty(M, d, device=DEV); bias = torch.randn(d, device=DEV); ot = torch.empty(M, 3 * d, device=DEV, dtype=dt)
x = torch.randn(M, d, device=DEV); g = torch.ones(d, device=DEV); b = torch.zeros(d, device=DEV); y = torch.empty(M, d, device=DEV); yt = torch.empty(M, d, device=DEV, dtype=dt)
def gemm(i=0): ops.gemm(A, Ws[i % 10], M, d, d, bias=bias, out_f32=o32)
def gemm128(): ops.gemm(A, W3, M, 3 * d, d, out_t=ot, tile=3)
def ln(): ops.layernorm(x, g, b, M, d, y_f32=y, y_t=yt, dtype=ops.code_of(yt))
tg = timeit(gemm); tl = timeit(ln); tq = timeit(gemm128)
print(f"same kernel repeated: gemm64 {tg:.2f} us, layernorm {tl:.2f} us, gemm128 (QKV) {tq:.2f} us")
def rot():
    for i in range(10): gemm(i)
print(f"gemm64 with 10 rotating weight matrices: {timeit(rot, n_rec=2) / 10:.2f} us per gemm")
def alt():
    gemm(); ln()
print(f"alternating gemm64, layernorm: {timeit(alt):.2f} us per pair (sum of separate: {tg + tl:.2f})")
def alt3():
    gemm128(); ln(); gemm(); ln()
print(f"alternating gemm128, ln, gemm64, ln: {timeit(alt3):.2f} us per group (sum of separate: {tq + tg + 2 * tl:.2f})")
# producer -> consumer: A is rewritten by a cast kernel right before every GEMM (as in the step, where the previous
# kernel's epilogue produces the operand)
xa = torch.randn(M, d, device=DEV)
def cast(): ops.cast(xa, A)
tc = timeit(cast)
def pc():
    cast(); gemm()
tpc = timeit(pc)
print(f"cast alone {tc:.2f} us; cast -> gemm64 reading the fresh A: {tpc:.2f} us per pair => gemm {tpc - tc:.2f} us (vs {tg:.2f} with a static A)")
o2 = torch.empty(M, d, device=DEV)
def chain():
    ops.gemm(A, Ws[0], M, d, d, bias=bias, out_f32=o32, out_t=yt); ops.gemm(yt, Ws[1], M, d, d, bias=bias, out_f32=o2, out_t=A)
print(f"gemm -> gemm chained through their bf16 outputs: {timeit(chain) / 2:.2f} us per gemm")
