"""Micro-benchmark of the step's kernels at cfg2 shapes (M = 800, d = 1024) -- run on the GPU box.
Each op is recorded into a Program 20x and replayed as a hipGraph; time = HIP events / launches."""
import os, sys, math, torch
sys.path.insert(0, 'face-diffusion-model_amd')
from fdm_amd import ops
from fdm_amd._lib import *
DEV = 'cuda:0'
torch.manual_seed(0)

def timeit(fn, n_rec=20, reps=20):
    prog = ops.Program()
    with prog:
        for _ in range(n_rec):
            fn()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        prog.instantiate()
        prog.replay(3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        prog.replay(reps)
        e1.record(s)
    s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n_rec * reps)

def main():
    dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
    M, d, ffn = (int(sys.argv[2]) if len(sys.argv) > 2 else 800), 1024, 2048
    shapes = [('fixed-overhead K=64', M, d, 64), ('enc/dec/out-proj', M, d, d), ('qkv', M, 3 * d, d), ('ffn1', M, ffn, d), ('ffn2', M, d, ffn)]
    for name, m, n, k in shapes:
        A = torch.randn(m, k, device=DEV).to(dt); W = (torch.randn(n, k, device=DEV) / math.sqrt(k)).to(dt)
        bias = torch.randn(n, device=DEV); res = torch.randn(m, n, device=DEV)
        o32 = torch.empty(m, n, device=DEV); ot = torch.empty(m, n, device=DEV, dtype=dt)
        us = timeit(lambda: ops.gemm(A, W, m, n, k, bias=bias, resid=res, out_f32=o32))
        fl = 2.0 * m * n * k
        print(f"gemm {name:18s} M={m} N={n} K={k}: {us:7.2f} us  {fl / us / 1e6:7.1f} TFLOP/s")
    if len(sys.argv) > 2: return
    B, H, L, hd = 4, 8, 200, 128
    Lpad = 224
    qkv = torch.randn(M, d, device=DEV).to(dt)
    kp = torch.randn(B * H, Lpad * hd, device=DEV).to(dt); vp = torch.randn(B * H, Lpad * hd, device=DEV).to(dt)
    o = torch.empty(M, d, device=DEV, dtype=dt); sl = torch.tensor([2.0 ** -(i + 1) for i in range(H)], device=DEV)
    us = timeit(lambda: ops.attention(qkv, kp, vp, o, B=B, H=H, L=L, hd=hd, ldq=d, ldo=d, Lpad=Lpad,
                                      scale=1 / math.sqrt(hd), causal=True, slopes=sl, period=30))
    print(f"attention B={B} H={H} L={L}: {us:7.2f} us")
    x = torch.randn(M, d, device=DEV); g = torch.ones(d, device=DEV); b = torch.zeros(d, device=DEV)
    y = torch.empty(M, d, device=DEV); yt = torch.empty(M, d, device=DEV, dtype=dt)
    us = timeit(lambda: ops.layernorm(x, g, b, M, d, y_f32=y, y_t=yt, dtype=ops.code_of(yt)))
    print(f"layernorm M={M}: {us:7.2f} us")
    n = M * d
    one = torch.ones(1000, device=DEV); st = torch.zeros(1, dtype=torch.int32, device=DEV); ts = torch.zeros(8, dtype=torch.int32, device=DEV) + 5
    us = timeit(lambda: ops.sched_step(0, x, y, y, n, n_per_clip=n // 4, tseq=ts, step=st, c1=one, c2=one, sigma=one, seed=1))
    print(f"sched (philox) n={n}: {us:7.2f} us")
    us = timeit(lambda: ops.cast(x, yt))
    print(f"cast n={n}: {us:7.2f} us")

if __name__ == "__main__":
    main()


def fold_costs():
    """Cost of the LayerNorm-folding epilogue options (bf16, M=800)."""
    import math
    dt = torch.bfloat16
    M, d = 800, 1024
    A = torch.randn(M, d, device=DEV).to(dt); W3 = (torch.randn(3 * d, d, device=DEV) / 32).to(dt); W1 = (torch.randn(d, 2 * d, device=DEV) / 45).to(dt)
    U = torch.randn(M, 2 * d, device=DEV).to(dt)
    b3 = torch.randn(3 * d, device=DEV); b1 = torch.randn(d, device=DEV); res = torch.randn(M, d, device=DEV)
    qkv = torch.empty(M, 3 * d, device=DEV, dtype=dt)
    kp = torch.zeros(32, 128 * 224, device=DEV, dtype=dt); vp = torch.zeros(32, 128 * 224, device=DEV, dtype=dt)
    stats = torch.rand(16, M, 2, device=DEV) * 1000 + 2000; cs = torch.randn(3 * d, device=DEV)
    o32 = torch.empty(M, d, device=DEV); ot = torch.empty(M, d, device=DEV, dtype=dt)
    gam = torch.ones(d, device=DEV); bet = torch.zeros(d, device=DEV)
    kw = dict(out_t=qkv, ldo_t=3 * d, out_kp=kp, kp_col0=d, out_vp=vp, vp_col0=2 * d, kv_L=200, kv_Lpad=224, kv_hd=128)
    print(f"qkv plain (packed K/V)        {timeit(lambda: ops.gemm(A, W3, M, 3 * d, d, bias=b3, **kw)):7.2f} us")
    print(f"qkv row-major only            {timeit(lambda: ops.gemm(A, W3, M, 3 * d, d, bias=b3, out_t=qkv, ldo_t=3 * d)):7.2f} us")
    print(f"qkv + rowstats + colsum       {timeit(lambda: ops.gemm(A, W3, M, 3 * d, d, bias=b3, ln_stat_in=stats, ln_nparts=16, ln_dim=d, ln_colsum=cs, **kw)):7.2f} us")
    print(f"ffn2 plain                    {timeit(lambda: ops.gemm(U, W1, M, d, 2 * d, bias=b1, resid=res, out_f32=o32)):7.2f} us")
    print(f"ffn2 + out_t                  {timeit(lambda: ops.gemm(U, W1, M, d, 2 * d, bias=b1, resid=res, out_f32=o32, out_t=ot)):7.2f} us")
    print(f"ffn2 + out_t + stat_out       {timeit(lambda: ops.gemm(U, W1, M, d, 2 * d, bias=b1, resid=res, out_f32=o32, out_t=ot, stat_out=stats)):7.2f} us")
    Wo = (torch.randn(d, d, device=DEV) / 32).to(dt)
    print(f"out-proj plain                {timeit(lambda: ops.gemm(A, Wo, M, d, d, bias=b1, resid=res, out_f32=o32)):7.2f} us")
    print(f"out-proj + rln                {timeit(lambda: ops.gemm(A, Wo, M, d, d, bias=b1, resid=res, out_f32=o32, ln_stat_in=stats, ln_nparts=16, ln_dim=d, rln_gamma=gam, rln_beta=bet)):7.2f} us")


if __name__ == '__main__' and len(sys.argv) > 3 and sys.argv[3] == 'fold':
    fold_costs()
