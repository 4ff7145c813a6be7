"""A test set's clips of different durations: one sampling call per group of clips (padded at the end; exact, the denoiser's
attention is causal) against the reference's loop of one B = 1 call per clip (samples/sample_diffusion_vocaset.py:51: bs = 1).

    python tools/bench_testset.py [--dtype bf16] [--clips 16] [--group 8] [--ddim 100]

Synthetic audio-encoder features, lengths uniform in [60, 240] latent frames (2-8 s at 30 fps).  Frames/s counts VALID frames.
Per-clip results of the two ways are compared bit for bit.  Prints one JSON line."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-diffusion-model_amd")]
import torch
from fdm_amd._lib import DTYPE_NAMES
from fdm_amd.denoiser import DenoiserPlan
from fdm_amd import presets, synth as W

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--clips", type=int, default=16)
ap.add_argument("--group", type=int, default=8)
ap.add_argument("--ddim", type=int, default=100)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
DEV = "cuda:0"
p = presets.get("vocaset")
g = torch.Generator().manual_seed(7)
Ls = torch.randint(60, 241, (a.clips,), generator=g).tolist()
hubs = [torch.randn(1, L, 1024, generator=g).to(DEV) for L in Ls]
xs = [torch.randn(1, L * p.G, p.c, generator=g).to(DEV) for L in Ls]
styles = torch.eye(p.n_style)[torch.randint(0, p.n_style, (a.clips,), generator=g)].to(DEV)
plan = DenoiserPlan("vocaset", W.make_fdm_weights("vocaset"), DTYPE_NAMES[a.dtype], DEV)


def sequential():
    out = []
    for b, L in enumerate(Ls):
        plan.prepare(hubs[b], styles[b:b + 1], L=L)
        out.append(plan.sample_ddim(xs[b], a.ddim))
    return out


def batched():
    out = [None] * a.clips
    order = sorted(range(a.clips), key=lambda b: Ls[b])
    for g0 in range(0, a.clips, a.group):
        grp = order[g0:g0 + a.group]
        Lmax = max(Ls[b] for b in grp)
        hub = torch.zeros(len(grp), Lmax, 1024, device=DEV)
        x = torch.zeros(len(grp), Lmax * p.G, p.c, device=DEV)
        for i, b in enumerate(grp):
            hub[i, :Ls[b]] = hubs[b][0]
            x[i, :Ls[b] * p.G] = xs[b][0]
        plan.prepare(hub, styles[grp], L=Lmax)
        lat = plan.sample_ddim(x, a.ddim)
        for i, b in enumerate(grp):
            out[b] = lat[i:i + 1, :Ls[b] * p.G]
    return out


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.reps, out


ts, outs = timed(sequential)
tb, outb = timed(batched)
same = all(torch.equal(x, y) for x, y in zip(outs, outb))
frames = sum(Ls)
print(json.dumps({"what": f"{a.clips} clips of {min(Ls)}..{max(Ls)} latent frames ({frames} in all), DDIM {a.ddim}, {a.dtype}", "group": a.group,
                  "batched_frames_per_s": round(frames / tb, 1), "sequential_b1_frames_per_s": round(frames / ts, 1), "speedup": round(ts / tb, 2),
                  "padded_rows_share": round(1 - frames / sum(max(Ls[b] for b in sorted(range(a.clips), key=lambda b: Ls[b])[g0:g0 + a.group]) * len(sorted(range(a.clips), key=lambda b: Ls[b])[g0:g0 + a.group]) for g0 in range(0, a.clips, a.group)), 3),
                  "per_clip_bit_identical": same}))
