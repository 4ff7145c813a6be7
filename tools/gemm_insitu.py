"""In-situ GEMM timing: each of the step's four GEMM shapes at M rows, as it runs inside the step -- a producer kernel
(LayerNorm writing the A operand, so A starts outside the consuming XCD's L2) followed by the GEMM, for 8 layers with
distinct weights (W streams from beyond L2), recorded as one program and replayed as a hipGraph.  The producer-only program
is timed too and subtracted.  usage: gemm_insitu.py [dtype] [M] [tiles...]"""
import sys
sys.path.insert(0, 'face-diffusion-model_amd')
import torch
from fdm_amd import ops
from fdm_amd._lib import DTYPE_NAMES, F32, ACT_RELU, ACT_NONE
DEV = 'cuda:0'
dt = DTYPE_NAMES[sys.argv[1]] if len(sys.argv) > 1 else 1
M = int(sys.argv[2]) if len(sys.argv) > 2 else 800
tiles = [int(t) for t in sys.argv[3:]] or list(range(0, 10))
shapes = {"out N1024 K1024": (1024, 1024, ACT_NONE), "qkv N3072 K1024": (3072, 1024, ACT_NONE), "ffn1 N2048 K1024": (2048, 1024, ACT_RELU),
          "ffn2 N1024 K2048": (1024, 2048, ACT_NONE)}
g = torch.Generator().manual_seed(0)
NL = 8


def mk(rows, cols):
    x = (torch.randn(rows, cols, generator=g) * 0.05).to(DEV)
    return ops.to_operand(x, dt) if dt != F32 else x


def timed(prog, reps=20):
    prog.instantiate()
    prog.replay(3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); prog.replay(reps); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3 / NL      # us per layer


for name, (N, K, act) in shapes.items():
    Ws = [mk(N, K) for _ in range(NL)]
    bias = torch.zeros(N, device=DEV)
    x32 = torch.randn(M, K, generator=g).to(DEV)
    gam, bet = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    A = ops.Split.empty(M, K, dt, DEV) if ops.is_split(dt) else torch.zeros(M, K, device=DEV, dtype=ops.tdtype(dt))
    out = ops.Split.empty(M, N, dt, DEV) if ops.is_split(dt) else torch.zeros(M, N, device=DEV, dtype=ops.tdtype(dt))
    rows = M * K // 1024

    def producer():
        # LayerNorm over the operand's memory viewed as rows of 1024 (the kernel's widest row): writes all of A
        ops.layernorm(x32, gam[:1024], bet[:1024], rows, 1024, y_t=A, dtype=dt)
    base = ops.Program()
    with base:
        for l in range(NL):
            producer()
    t_base = timed(base)
    res = []
    for tile in tiles:
        prog = ops.Program()
        with prog:
            for l in range(NL):
                producer()
                ops.gemm(A, Ws[l], M, N, K, bias=bias, act=act, out_t=out, tile=tile)
        res.append((timed(prog) - t_base, tile))
    print(f"{name} M={M}: producer {t_base:.2f} us; GEMM us by tile: " + "  ".join(f"{t}:{v:.2f}" for v, t in res) + f"   best tile {min(res)[1]}")
