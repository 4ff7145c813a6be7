"""[needs the library of commit 2451829: the fused layer tail was measured 9-20 % slower on every shape and removed again -- profiles/README.md,
round 4, profiles/r4_fused_tail/]
A/B of the fused layer tail (fdm_plan_set "fuse_tail": out-proj .. norm3 of every layer as one XCD-resident launch, csrc/tail.hpp)
against the per-operator step program, on one box, alternating; also checks the two programs produce the same bits.
    python tools/bench_tail.py [shape ...]        shapes: cfg1 cfg2 cfg3 cfg4 cfg5rows mead249 voc8x498"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "face-diffusion-model_amd"))
from fdm_amd import synth as W  # noqa: E402
from fdm_amd._lib import DTYPE_NAMES  # noqa: E402
from fdm_amd.denoiser import DenoiserPlan  # noqa: E402

DEV = "cuda:0"
SHAPES = {"cfg1": ("vocaset", 1, 100, False), "cfg2": ("vocaset", 4, 200, False), "cfg3": ("mead", 4, 300, True),
          "cfg4": ("biwi", 4, 200, False), "cfg5rows": ("vocaset", 4, 498, False), "mead249": ("mead", 1, 249, False),
          "voc8x498": ("vocaset", 8, 498, False)}


def run(name, dt):
    preset, B, L, cfg = SHAPES[name]
    w = W.make_fdm_weights(preset)
    inp = W.synth_inputs(preset, B, L, seed=1)
    hub = inp["hub"][:, :, :768].contiguous() if preset == "biwi" else inp["hub"]
    plan = DenoiserPlan(preset, w, DTYPE_NAMES[dt], DEV)
    xT = inp["x"].to(DEV)
    ts = list(range(999, 799, -1))
    outs, times, launches = {}, {0: [], 1: []}, {}
    kw = dict(cfg_scale=2.5) if cfg else {}
    for rnd in range(3):
        for fuse in (0, 1):
            plan.set("fuse_tail", fuse)
            plan.prepare(hub, inp["style"], inp.get("emo"), L=L, cfg=cfg)
            o = plan.sample_ddpm(xT, ts, seed=5, **kw)        # records + warms
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                o = plan.sample_ddpm(xT, ts, seed=5, **kw)
            torch.cuda.synchronize()
            times[fuse].append(round((time.perf_counter() - t0) / 3 / len(ts) * 1e3, 4))
            outs[fuse] = o
            launches[fuse] = plan.get("launches_per_step")
    err = plan.get("tail_errors")
    same = bool(torch.equal(outs[0], outs[1]))
    a, b = min(times[0]), min(times[1])
    print(f"{name:9s} {dt:6s} rows {B * L * (2 if cfg else 1):5d}: per-operator {a:.4f} ms/step ({launches[0]} launches) | fused tail {b:.4f} ms/step "
          f"({launches[1]} launches, {(a / b - 1) * 100:+.1f} % steps/s) | bit-identical {same} | finite {bool(torch.isfinite(outs[1]).all())} | "
          f"spin time-outs {err} | rounds {times}", flush=True)


if __name__ == "__main__":
    names = sys.argv[1:] or ["cfg2"]
    for n in names:
        for dt in ("bf16", "f16x3"):
            run(n, dt)
