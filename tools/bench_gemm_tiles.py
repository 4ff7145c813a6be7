"""Isolated GEMM throughput per output tile at a given shape (run on the GPU box).

    python tools/bench_gemm_tiles.py [bf16|f16x3|f32] M N K [M N K ...]

Each (shape, tile) pair is recorded 8x over DISTINCT weight matrices (so the weights come from beyond L2, as they do
inside the step) into a Program, replayed as a hipGraph; time = HIP events / launches.  Prints one line per (shape, tile)
and the fraction of the dense 16-bit MFMA peak (2.5 PFLOP/s; fp32: 157.3 TFLOP/s)."""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "face-diffusion-model_amd"))
from fdm_amd import ops  # noqa: E402
from fdm_amd._lib import DTYPE_NAMES, F32  # noqa: E402

DEV = "cuda:0"
TILES = {0: "auto", 1: "64x64", 2: "128x64", 3: "128x128", 8: "64x64_s2", 9: "32x64_s3", 10: "256x128_pp", 11: "80x128", 12: "64x128",
         0x201: "64x64 lockstep", 0x202: "128x64 lockstep"}      # (0x200 = FDM_TILE_LOCKSTEP: the loop without loader waves)


def timeit(fn, n_rec=8, reps=10):
    prog = ops.Program()
    with prog:
        for i in range(n_rec):
            fn(i)
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
    kind = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    code = DTYPE_NAMES[kind]
    dims = [int(v) for v in sys.argv[2:]] or [6400, 1024, 2048]
    peak = 157.3 if kind == "f32" else 2500.0
    tiles = dict(TILES)
    torch.manual_seed(0)
    out = []
    for i in range(0, len(dims), 3):
        m, n, k = dims[i:i + 3]
        A32 = torch.randn(m, k, device=DEV)
        A = ops.to_operand(A32, code) if code != F32 else A32
        Ws = []
        for _ in range(8):
            w32 = torch.randn(n, k, device=DEV) / math.sqrt(k)
            Ws.append(ops.to_operand(w32, code) if code != F32 else w32)
        bias = torch.randn(n, device=DEV)
        res = torch.randn(m, n, device=DEV)
        o32 = torch.empty(m, n, device=DEV)
        for tile, name in tiles.items():
            try:
                us = timeit(lambda j: ops.gemm(A, Ws[j], m, n, k, bias=bias, resid=res, out_f32=o32, tile=tile))
            except Exception as e:  # a tile the kind does not instantiate
                print(f"gemm {kind} M={m} N={n} K={k} tile {name}: unsupported ({str(e)[:60]})")
                continue
            tf = 2.0 * m * n * k / us / 1e6
            print(f"gemm {kind} M={m} N={n} K={k} tile {name:10s}: {us:8.2f} us  {tf:7.1f} TFLOP/s  {tf / peak * 100:5.1f} % of peak", flush=True)
            out.append({"kind": kind, "M": m, "N": n, "K": k, "tile": name, "us": round(us, 3), "tflops": round(tf, 1), "frac": round(tf / peak, 4)})
    dst = os.environ.get("FDM_GEMM_TILES_JSON")
    if dst:
        json.dump(out, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main()
