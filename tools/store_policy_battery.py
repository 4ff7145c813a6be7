"""One deterministic pass over every kernel family of the path, printing a sha256 per output (tests/test_store_policy_gpu.py runs it
under the product library and under the FDM_PLAIN_STORES build -- FDM_LIB_PATH -- and compares the lines): the step program in every
arithmetic mode (DDPM with injected and Philox noise, DDIM, CFG, condition batching, K slices, the norm3 fold), the audio encoders, the
VQ encoder / quantiser / decoder, resampling, and the metric kernels."""
import hashlib
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-diffusion-model_amd")]
import torch
from fdm_amd._lib import BF16, F16, F16X3, F32
from fdm_amd.denoiser import DenoiserPlan
from fdm_amd.hubert import WAV2VEC2_BASE, HubertPlan
from fdm_amd.vq import VQPlan
from fdm_amd import synth as W

DEV = "cuda:0"


def show(name, t):
    t = t.detach().contiguous().cpu()
    print(f"{name} {tuple(t.shape)} {hashlib.sha256(t.view(torch.uint8).numpy().tobytes()).hexdigest()[:24]}", flush=True)


gen = torch.Generator().manual_seed(1)
for preset, B, L, cfg in (("vocaset", 2, 37, False), ("mead", 2, 40, True), ("biwi", 1, 24, False)):
    w = W.make_fdm_weights(preset)
    inp = W.synth_inputs(preset, B, L, seed=5)
    if preset == "biwi":      # wav2vec2-base features: two 768-wide frames per latent frame
        inp["hub"] = torch.randn(B, 2 * L, 768, generator=torch.Generator().manual_seed(3))
    # (VOCASET in every mode; MEAD + CFG and BIWI -- head_dim 256 -- in the 16-bit modes whose kernels differ from VOCASET's)
    modes = {"vocaset": ((F32, "f32"), (BF16, "bf16"), (F16X3, "f16x3"), (F16, "f16")), "mead": ((BF16, "bf16"), (F16X3, "f16x3")), "biwi": ((BF16, "bf16"), (F16X3, "f16x3"))}[preset]
    for dt, nm in modes:
        plan = DenoiserPlan(preset, w, dt, DEV)
        plan.prepare(inp["hub"], inp["style"], inp.get("emo"), L=L, cfg=cfg)
        x = inp["x"].to(DEV)
        noise = torch.randn(4, *inp["x"].shape, generator=torch.Generator().manual_seed(3))
        show(f"{preset} {nm} denoise", plan.denoise(x, 321))
        show(f"{preset} {nm} ddpm injected", plan.sample_ddpm(x, [9, 5, 1, 0], noise=noise))
        show(f"{preset} {nm} ddpm philox", plan.sample_ddpm(x, list(range(19, -1, -1)), seed=7))
        show(f"{preset} {nm} ddim", plan.sample_ddim(x, 10))
        if dt in (BF16, F16X3) and not cfg:
            plan.set("ksplit.out", 2); plan.set("ksplit.ffn2", 4)
            plan.prepare(inp["hub"], inp["style"], inp.get("emo"), L=L, cfg=cfg)
            show(f"{preset} {nm} ddim K slices", plan.sample_ddim(x, 6))
            plan.set("ksplit.out", 1); plan.set("ksplit.ffn2", 1)
            plan.set("fuse_ln3", 1)
            plan.prepare(inp["hub"], inp["style"], inp.get("emo"), L=L, cfg=cfg)
            show(f"{preset} {nm} ddim norm3 folded", plan.sample_ddim(x, 6))
wav = torch.randn(2, 16000, generator=gen) * 0.1
for dt, nm in ((F32, "f32"), (BF16, "bf16"), (F16X3, "f16x3")):
    show(f"hubert-large 2 layers {nm}", HubertPlan(W.make_hubert_weights(2), 2, dt, DEV).forward(wav)[0])
    if dt != F32:
        show(f"wav2vec2-base 2 layers {nm}", HubertPlan(W.make_wav2vec_weights(2), 2, dt, DEV, cfg=WAV2VEC2_BASE).forward(wav)[0])
from fdm_amd import presets
for preset in ("vocaset", "mead", "biwi"):
    p = presets.get(preset)
    L = 9
    wv = W.make_vq_weights(preset, encoder=True)
    emo = torch.eye(7)[3].unsqueeze(0).expand(2, -1) if p.n_books > 1 else None
    for dt, nm in {"vocaset": ((F32, "f32"), (BF16, "bf16"), (F16X3, "f16x3")), "mead": ((F32, "f32"),), "biwi": ((BF16, "bf16"),)}[preset]:
        vq = VQPlan(preset, wv, dt, DEV)
        verts = torch.randn(2, L, p.V3, generator=gen) * 0.01
        z = vq.encode(verts.to(DEV), emo)
        show(f"vq encode {preset} {nm}", z)
        zq, loss, (perp, me, idx) = vq.quant_full(z * 0.2, emo)
        show(f"vq quant {preset} {nm} zq", zq); show(f"vq quant {preset} {nm} idx", idx); show(f"vq quant {preset} {nm} stats", torch.stack([loss, perp]))
        show(f"vq decode {preset} {nm}", vq.decode(zq))
from fdm_amd import metrics
gt, pr = torch.randn(11, 300, 3, generator=gen), torch.randn(11, 300, 3, generator=gen)
ve = metrics.vertex_error(gt, pr, list(range(0, 300, 3)), device=DEV)
show("vertex_error", torch.cat([ve["frame_max"].double().cpu(), torch.tensor([ve["max"], ve["mean_sq"], ve["mean_dist"]], dtype=torch.float64)]))
show("motion_std", torch.tensor([metrics.motion_std(gt, torch.zeros(300, 3), list(range(0, 300, 2)), device=DEV)], dtype=torch.float64))
