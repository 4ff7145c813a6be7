"""Round 5: the loader-wave form of the GEMM tiles (default for the 16-bit kinds) against the lockstep k loop (FDM_TILE_LOCKSTEP), isolated,
8 distinct weights per shape, graph replay; bit-identity checked.   python tools/bench_ldw.py [bf16|f16x3|f32] [rows]"""
import math
import sys

import torch

sys.path.insert(0, 'face-diffusion-model_amd'); sys.path.insert(0, 'tools')
from fdm_amd import ops
from fdm_amd._lib import *  # noqa: F401,F403

DEV = 'cuda:0'
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
M = int(sys.argv[2]) if len(sys.argv) > 2 else 800
code = {'bf16': BF16, 'f16x3': F16X3, 'f32': F32}[mode]
NAMES = {1: "64x64", 8: "64x64/2", 9: "32x64", 11: "80x128", 2: "128x64", 12: "64x128", 3: "128x128"}
LOCKSTEP = 0x200
torch.manual_seed(0)


def timeit(fn, n_rec=8, reps=10):
    prog = ops.Program()
    with prog:
        for i in range(n_rec):
            fn(i)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        prog.instantiate(); prog.replay(3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); prog.replay(reps); e1.record(s)
    s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n_rec * reps)


print(f"== {mode}, {M} rows: us lockstep -> loader waves ==")
for (n, k, relu, tiles) in ((1024, 1024, False, (1, 9, 8)), (1024, 2048, False, (1, 9, 8)), (2048, 1024, True, (1, 2, 12, 3)), (3072, 1024, False, (11, 2, 3))):
    A = ops.to_operand(torch.randn(M, k, device=DEV), code)
    Ws = [ops.to_operand(torch.randn(n, k, device=DEV) / math.sqrt(k), code) for _ in range(8)]
    bias = torch.randn(n, device=DEV); res = torch.randn(M, n, device=DEV)
    o32 = torch.empty(M, n, device=DEV)
    ot = None if code == F32 else (ops.Split.empty(M, n, code, DEV) if ops.is_split(code) else torch.empty(M, n, device=DEV, dtype=torch.bfloat16))
    kw = dict(bias=bias, act=ACT_RELU if relu else ACT_NONE)
    kw.update(dict(out_t=ot) if (relu and ot is not None) else dict(resid=res, out_f32=o32))
    line = f"{M} x {n} x {k}{' relu->operand' if relu else ''}:"
    first = None
    for tile in tiles:
        t = []
        for flag in (LOCKSTEP, 0):
            ops.gemm(A, Ws[0], M, n, k, tile=tile | flag, **kw); torch.cuda.synchronize()
            cur = (ot.planes if ops.is_split(code) else ot).clone() if (relu and ot is not None) else o32.clone()
            if first is None: first = cur
            ok = torch.equal(cur, first)
            t.append(timeit(lambda j: ops.gemm(A, Ws[j], M, n, k, tile=tile | flag, **kw)))
        line += f"  {NAMES[tile]} {t[0]:6.2f} -> {t[1]:6.2f}{'' if ok else ' MISMATCH'}"
    print(line, flush=True)
