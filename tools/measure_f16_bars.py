"""Single-plane fp16 step program (FDM_F16, round 6) beside bf16: measured max-abs distances at the places the bf16 bars of
tests/test_denoiser_gpu.py / tests/test_configs_gpu.py are stated (run on the GPU box; the F16 bars in tests/ are 2x these), and the
largest magnitude any operand copy takes (fp16 clamps at 65504)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-diffusion-model_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
from fdm_amd._lib import BF16, F16, F16X3, F32
from fdm_amd.denoiser import DenoiserPlan
from oracle import fdm_oracle as FO, weights as W

DEV = "cuda:0"
mad = lambda a, b: float((torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max())
NAME = {BF16: "bf16", F16: "f16", F16X3: "f16x3", F32: "f32"}
MODES = [BF16, F16]

# 1. single denoiser calls vs the reference goldens (tests/test_denoiser_gpu.py::test_single_step_bf16_stated_tolerance)
for preset in ("vocaset", "mead"):
    g = np.load(os.path.join(ROOT, f"tests/golden/fdm_step_{preset}.npz"))
    w = W.make_fdm_weights(preset)
    for dt in MODES:
        plan = DenoiserPlan(preset, w, dt, DEV)
        worst = 0.0
        for (L, t) in g["cases"].tolist():
            inp = W.synth_inputs(preset, 1, L, seed=100 + L)
            plan.prepare(inp["hub"], inp["style"], inp.get("emo"), L=L)
            worst = max(worst, mad(plan.denoise(inp["x"].to(DEV), t)[0], g[f"x0_L{L}_t{t}"]))
        print(f"single step vs reference goldens, {preset}, {NAME[dt]}: {worst:.3e}", flush=True)

# 2. cfg2 at its benched shape, un-attenuated: last three DDPM steps + denoise(t = 500) vs the oracle; 50-step DDIM chain and the
#    full 1000-step chain vs the fp32 program
preset, B, L = "vocaset", 4, 200
w = W.make_fdm_weights(preset)
inp = W.synth_inputs(preset, B, L, seed=2)
ts3 = [2, 1, 0]
noise = torch.randn(3, *inp["x"].shape, generator=torch.Generator().manual_seed(0))
refs = {}
for b in (0, 3):
    den = lambda x, t: FO.fdm_forward(w, preset, inp["hub"][b:b + 1], t, x, inp["style"][b:b + 1], None, folded=True)
    ref = []
    FO.p_sample_loop(den, inp["x"][b:b + 1].clone(), noise[:, b:b + 1], ts3, record=ref)
    refs[b] = (torch.stack(ref), den(inp["x"][b:b + 1], 500))
full = {}
for dt in [F32] + MODES:
    plan = DenoiserPlan(preset, w, dt, DEV)
    plan.prepare(inp["hub"], inp["style"], L=L)
    rec = []
    plan.sample_ddpm(inp["x"].to(DEV), ts3, noise=noise, record=rec)
    x0 = plan.denoise(inp["x"].to(DEV), 500)
    worst = max(mad(torch.stack(rec)[:, b:b + 1], refs[b][0]) for b in (0, 3))
    worst_d = max(mad(x0[b:b + 1], refs[b][1]) for b in (0, 3))
    full[dt] = (plan.sample_ddpm(inp["x"].to(DEV), list(range(999, -1, -1)), seed=5), plan.sample_ddim(inp["x"].to(DEV), 50))
    print(f"cfg2 4 x 200, {NAME[dt]}: last 3 DDPM steps vs oracle {worst:.3e}; denoise(t = 500) vs oracle {worst_d:.3e}", flush=True)
for dt in MODES:
    print(f"cfg2 4 x 200, {NAME[dt]}: after the 1000-step DDPM chain vs the fp32 program {mad(full[dt][0], full[F32][0]):.3e}; after DDIM 50 "
          f"{mad(full[dt][1], full[F32][1]):.3e} (latent max {float(full[F32][0].abs().max()):.2f})", flush=True)

# 3. cfg3 (MEAD + CFG) last three steps, clip 1
preset, B, L = "mead", 4, 300
w = W.make_fdm_weights(preset)
inp = W.synth_inputs(preset, B, L, seed=3)
noise = torch.randn(3, *inp["x"].shape, generator=torch.Generator().manual_seed(0))
den = lambda x, t: FO.fdm_forward_cfg(w, preset, inp["hub"][1:2], t, x, inp["style"][1:2], inp["emo"][1:2], 2.5, folded=True)
ref = []
FO.p_sample_loop(den, inp["x"][1:2].clone(), noise[:, 1:2], ts3, record=ref)
for dt in MODES:
    plan = DenoiserPlan(preset, w, dt, DEV)
    plan.prepare(inp["hub"], inp["style"], inp["emo"], L=L, cfg=True)
    rec = []
    plan.sample_ddpm(inp["x"].to(DEV), ts3, noise=noise, cfg_scale=2.5, record=rec)
    print(f"cfg3 4 x 300 + CFG, {NAME[dt]}: last 3 steps vs oracle {mad(torch.stack(rec)[:, 1:2], torch.stack(ref)):.3e}", flush=True)

# 4. chains vs the reference's own chain goldens (what bench.py's parity leg reads)
import bench
w = W.make_fdm_weights("vocaset")
for dt in MODES + [F16X3]:
    print(f"chains_vocaset goldens (bench.py's parity leg), {NAME[dt]}: {bench.parity_vs_reference(DenoiserPlan('vocaset', w, dt, DEV), DEV):.3e}", flush=True)
