"""Do small kernels on different HIP streams overlap on this box?  (eager launches, no graphs)"""
import sys, math, time, torch
sys.path.insert(0, 'face-diffusion-model_amd')
from fdm_amd import ops
DEV = 'cuda:0'
M, d = 200, 1024
dt = torch.bfloat16
def mk():
    A = torch.randn(M, d, device=DEV).to(dt); W = torch.randn(d, d, device=DEV).to(dt); o = torch.empty(M, d, device=DEV)
    return A, W, o
bufs = [mk() for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
def run(ns, iters=300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        for s in range(ns):
            with torch.cuda.stream(streams[s]):
                A, W, o = bufs[s]
                ops.gemm(A, W, M, d, d, out_f32=o)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (iters * ns) * 1e6
for ns in (1, 2, 4):
    run(ns, 50)
    print(f"{ns} stream(s): {run(ns):.2f} us per kernel (wall / total kernels)")
# big kernels: is there overlap at all?
M2 = 200
x = [torch.randn(4096, 4096, device=DEV) for _ in range(4)]
def run2(ns, iters=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(iters):
        for s in range(ns):
            with torch.cuda.stream(streams[s]):
                A, W, o = bufs[s]
                for _ in range(20):
                    ops.gemm(A, W, M, d, d, out_f32=o)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / (iters * ns * 20) * 1e6
for ns in (1, 2, 4):
    print(f"batched x20, {ns} stream(s): {run2(ns):.2f} us per kernel")
