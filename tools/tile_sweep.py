"""Tuner picks across row counts (how far the library's untuned tile heuristic is from the tuned set).
usage: FDM_TUNE_VERBOSE=1 python tools/tile_sweep.py 2> picks.txt"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "face-diffusion-model_amd"))
import torch
from fdm_amd import synth as W
from fdm_amd._lib import DTYPE_NAMES
from fdm_amd.denoiser import DenoiserPlan

dev = torch.device("cuda:0")
for preset, cfg in (("vocaset", False), ("mead", True), ("biwi", False)):
    w = W.make_fdm_weights(preset, seed=0)
    for dt in ("bf16", "f16x3"):
        plan = DenoiserPlan(preset, w, DTYPE_NAMES[dt], dev)
        for B, L in ((1, 100), (1, 200), (2, 200), (4, 150), (4, 200), (4, 300), (8, 200), (4, 498), (8, 300), (16, 200), (8, 498)):
            inp = W.synth_inputs(preset, B, L, seed=1)
            hub = inp["hub"][:, :, :768].contiguous() if preset == "biwi" else inp["hub"]
            plan.prepare(hub, inp["style"], inp.get("emo"), L=L, cfg=cfg)
            sys.stderr.write(f"## {preset} {dt} B={B} L={L} cfg={cfg}\n"); sys.stderr.flush()
            plan.tune()
            sys.stderr.write(f"   kept: { {k: v for k, v in plan.tiles.items() if v} }\n"); sys.stderr.flush()
