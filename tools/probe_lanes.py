"""Probe: the B clips of one sampling call split over K concurrent lanes (K plans, K streams, B/K clips each) against the one
B-clip chain.  usage: python tools/probe_lanes.py [dtype] [B] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "face-diffusion-model_amd"))
import torch
from fdm_amd import presets, synth as W
from fdm_amd._lib import DTYPE_NAMES
from fdm_amd.denoiser import DenoiserPlan

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
T = int(sys.argv[3]) if len(sys.argv) > 3 else 200
L = 200
dev = torch.device("cuda:0")
weights = W.make_fdm_weights("vocaset", seed=0)
inp = W.synth_inputs("vocaset", B, L, seed=1)
ts = list(range(T - 1, -1, -1))


def run(K):
    n = B // K
    plans, streams, xs = [], [], []
    for k in range(K):
        pl = DenoiserPlan("vocaset", weights, DTYPE_NAMES[dtype], dev)
        sl = slice(k * n, (k + 1) * n)
        pl.prepare(inp["hub"][sl], inp["style"][sl], None, L=L)
        pl.tune()
        plans.append(pl); streams.append(torch.cuda.Stream(dev)); xs.append(inp["x"][sl].to(dev))
    torch.cuda.synchronize()

    def call():
        outs = []
        for k in range(K):
            with torch.cuda.stream(streams[k]):
                outs.append(plans[k].sample_ddpm(xs[k], ts, seed=1234, clip0=k * n))
        torch.cuda.synchronize()
        return torch.cat(outs)
    call()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); out = call(); best = min(best, time.perf_counter() - t0)
    return best, out


ref = None
for K in (1, 2, 4, 1, 2, 4):
    if B % K:
        continue
    t, out = run(K)
    if ref is None:
        ref = out
    print(f"{dtype} B={B} lanes={K}: {t / T * 1e3:.4f} ms per diffusion step, {B * L * 1000 / T / t * (T / 1000):.1f} frames/s at 1000 steps, identical to 1 lane: {bool(torch.equal(out, ref))}", flush=True)
