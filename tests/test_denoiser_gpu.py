"""GPU: the HIP denoiser + scheduler step program against (a) the committed golden vectors produced by
the reference itself and (b) the CPU oracle on the same seeded inputs.

Tolerances: fp32 (parity) mode 1e-4 max-abs on O(4) latents -- the bar BASELINE.json states;
bf16 (throughput) mode 8e-2 max-abs per denoiser call (bf16 operands carry 8 significant bits)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fdm_amd.denoiser import DenoiserPlan  # noqa: E402
from fdm_amd._lib import BF16, F16, F16X3, F32  # noqa: E402
from oracle import fdm_oracle as FO  # noqa: E402
from oracle import weights as W  # noqa: E402

DEV = "cuda:0"
TOL32 = 1e-4
TOLBF = 8e-2
# single-plane fp16 step program (FDM_F16, round 6: the split kind's hi plane alone -- bf16's bytes and MFMA rate, 11 significand bits):
# 2x the distance measured on MI355X against the reference goldens (2.86e-3 VOCASET, 2.88e-3 MEAD; tools/measure_f16_bars.py,
# profiles/r6_f16/bars.txt) -- outside the 1e-4 contract like bf16, 8x closer to the reference at the same speed
TOLF16 = 6e-3
# the modes that meet the contract tolerance (BASELINE.json: 1e-4 max-abs vs the reference): exact fp32 MFMA, and split-fp16
# operands on the 16-bit matrix cores (three MFMA passes per product, 22 significant bits per operand)
PARITY_MODES = [F32, F16X3]


def mad(a, b):
    return float((torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max())


_PLANS = {}


def plan_for(preset, dtype):
    key = (preset, dtype)
    if key not in _PLANS:
        _PLANS[key] = (DenoiserPlan(preset, W.make_fdm_weights(preset), dtype, DEV), W.make_fdm_weights(preset))
    return _PLANS[key]


@pytest.mark.parametrize("dtype", PARITY_MODES)
@pytest.mark.parametrize("preset", ["vocaset_tiny", "mead_tiny", "vocaset", "mead"])
def test_single_step_vs_golden_fp32(golden, preset, dtype):
    g = golden(f"fdm_step_{preset}")
    plan, _ = plan_for(preset, dtype)
    for (L, t) in g["cases"].tolist():
        inp = W.synth_inputs(preset, 1, L, seed=100 + L)
        plan.prepare(inp["hub"], inp["style"], inp.get("emo"), L=L)
        out = plan.denoise(inp["x"].to(DEV), t)
        assert mad(out[0], g[f"x0_L{L}_t{t}"]) < TOL32, (preset, L, t)


@pytest.mark.parametrize("preset", ["vocaset", "mead"])
def test_single_step_bf16_stated_tolerance(golden, preset):
    g = golden(f"fdm_step_{preset}")
    plan, _ = plan_for(preset, BF16)
    for (L, t) in g["cases"].tolist():
        inp = W.synth_inputs(preset, 1, L, seed=100 + L)
        plan.prepare(inp["hub"], inp["style"], inp.get("emo"), L=L)
        out = plan.denoise(inp["x"].to(DEV), t)
        assert mad(out[0], g[f"x0_L{L}_t{t}"]) < TOLBF, (preset, L, t)


@pytest.mark.parametrize("preset", ["vocaset", "mead"])
def test_single_step_f16_stated_tolerance(golden, preset):
    """The single-plane fp16 mode against the reference's denoiser outputs: inside 6e-3 where bf16 states 8e-2 (and must stay an
    order of magnitude inside bf16's measured 2.3e-2 -- otherwise the mode has no reason to exist)."""
    g = golden(f"fdm_step_{preset}")
    plan, _ = plan_for(preset, F16)
    worst = 0.0
    for (L, t) in g["cases"].tolist():
        inp = W.synth_inputs(preset, 1, L, seed=100 + L)
        plan.prepare(inp["hub"], inp["style"], inp.get("emo"), L=L)
        worst = max(worst, mad(plan.denoise(inp["x"].to(DEV), t)[0], g[f"x0_L{L}_t{t}"]))
    print(f"[f16 {preset}] max-abs vs reference goldens {worst:.3e} (bar {TOLF16:.0e})")
    assert worst < TOLF16, preset


def test_f16x3_split_sits_at_the_fp32_kernels_distance(golden):
    """Why the split mode uses fp16 planes: 11 + 11 bits per operand put a full-size denoiser call at the fp32 MFMA kernel's own
    distance from the reference.  (bf16 planes -- 8 + 8 bits -- were built and measured in rounds 2-4: 4.9e-5 per call, > 1e-4
    over a 50-step chain; tools/sim_split_precision.py predicted both before any kernel was written.  That kind is no longer part
    of the library.)"""
    g = golden("fdm_step_vocaset")
    err = {}
    for dt in (F32, F16X3):
        plan, _ = plan_for("vocaset", dt)
        e = 0.0
        for (L, t) in g["cases"].tolist():
            inp = W.synth_inputs("vocaset", 1, L, seed=100 + L)
            plan.prepare(inp["hub"], inp["style"], L=L)
            e = max(e, mad(plan.denoise(inp["x"].to(DEV), t)[0], g[f"x0_L{L}_t{t}"]))
        err[dt] = e
    print(f"max-abs vs reference goldens: f32 {err[F32]:.2e}  f16x3 {err[F16X3]:.2e}")
    assert err[F16X3] < 3 * max(err[F32], 5e-6) and err[F16X3] < TOL32


@pytest.mark.parametrize("dtype", PARITY_MODES)
@pytest.mark.parametrize("preset", ["vocaset_tiny", "vocaset", "mead"])
def test_chains_vs_golden_fp32(golden, preset, dtype):
    """DDPM t=9..0 and 999..990 with injected noise (after every step), DDIM 3 and 50 steps."""
    g = golden(f"chains_{preset}")
    plan, _ = plan_for(preset, dtype)
    L = int(g["L"])
    inp = W.synth_inputs(preset, 1, L, seed=7)
    plan.prepare(inp["hub"], inp["style"], inp.get("emo"), L=L)
    for name in ("lo", "hi"):
        rec = []
        noise = torch.from_numpy(g[f"ddpm_{name}_noise"])
        plan.sample_ddpm(inp["x"].to(DEV), g[f"ddpm_{name}_t"].tolist(), noise=noise, record=rec)
        assert mad(torch.stack(rec), g[f"ddpm_{name}_steps"]) < TOL32, name
        # the captured hipGraph gives bit-identical results to eager launches
        a = plan.sample_ddpm(inp["x"].to(DEV), g[f"ddpm_{name}_t"].tolist(), noise=noise, use_graph=True)
        b = plan.sample_ddpm(inp["x"].to(DEV), g[f"ddpm_{name}_t"].tolist(), noise=noise, use_graph=False)
        assert torch.equal(a, b)
        assert mad(a, g[f"ddpm_{name}_steps"][-1]) < TOL32
    if "ddim_3_final" in g:
        for steps in (3, 50):
            out = plan.sample_ddim(inp["x"].to(DEV), steps)
            assert mad(out, g[f"ddim_{steps}_final"]) < TOL32, steps


@pytest.mark.parametrize("dtype", PARITY_MODES)
def test_cfg_two_pass_mix_vs_golden(golden, dtype):
    g = golden("cfg_mead")
    plan, _ = plan_for("mead", dtype)
    L, t = int(g["L"]), int(g["t"])
    inp = W.synth_inputs("mead", 1, L, seed=55)
    plan.prepare(inp["hub"], inp["style"], inp["emo"], L=L, cfg=True)
    out, unc = plan.denoise(inp["x"].to(DEV), t, cfg_scale=2.5, return_uncond=True)
    assert mad(out[0], g["mix"]) < TOL32
    assert mad(unc[0], g["uncond"]) < TOL32


@pytest.mark.parametrize("preset,dtype", [("vocaset", F32), ("mead", F32), ("vocaset", BF16), ("vocaset", F16X3), ("mead", F16X3)])
def test_batched_clips_equal_independent_b1_calls(preset, dtype):
    """B > 1 == independent B = 1 reference calls; results do not depend on batch composition."""
    plan, w = plan_for(preset, dtype)
    B, L, t = 3, 37, 640
    inp = W.synth_inputs(preset, B, L, seed=21)
    plan.prepare(inp["hub"], inp["style"], inp.get("emo"), L=L)
    out = plan.denoise(inp["x"].to(DEV), t).cpu()
    ref = FO.fdm_forward(w, preset, inp["hub"], t, inp["x"], inp["style"], inp.get("emo"), folded=True)
    assert mad(out, ref) < (TOLBF if dtype == BF16 else TOL32)
    for b in (0, 2):
        plan.prepare(inp["hub"][b:b + 1], inp["style"][b:b + 1], None if "emo" not in inp else inp["emo"][b:b + 1], L=L)
        one = plan.denoise(inp["x"][b:b + 1].to(DEV), t).cpu()
        assert torch.equal(one[0], out[b]), "per-clip result depends on the batch"


@pytest.mark.parametrize("preset,dtype,So,Sf", [("vocaset", F32, 2, 4), ("vocaset", F16X3, 4, 4), ("mead", F16X3, 2, 4), ("vocaset", BF16, 2, 4)])
def test_split_k_plan_property_keeps_parity_and_batch_independence(golden, preset, dtype, So, Sf):
    """fdm_plan_set "ksplit.out" / "ksplit.ffn2" (the single-clip callers' setting: K slices of the out-proj / FFN2 GEMMs, summed by
    the LayerNorm launch that follows): same bars against the reference goldens, the graph == eager launches, and -- S being a
    property of the plan, not of the shape -- a clip still computes the same bits in every batch composition."""
    w = W.make_fdm_weights(preset)
    plan = DenoiserPlan(preset, w, dtype, DEV)
    plan.set("ksplit.out", So); plan.set("ksplit.ffn2", Sf)
    assert plan.get("ksplit.out") == So and plan.get("ksplit.ffn2") == Sf
    g = golden(f"fdm_step_{preset}")
    worst = 0.0
    for (L, t) in g["cases"].tolist():
        inp = W.synth_inputs(preset, 1, L, seed=100 + L)
        plan.prepare(inp["hub"], inp["style"], inp.get("emo"), L=L)
        worst = max(worst, mad(plan.denoise(inp["x"].to(DEV), t)[0], g[f"x0_L{L}_t{t}"]))
    print(f"[{preset} dtype {dtype} ksplit {So}/{Sf}] single steps vs reference goldens: max-abs {worst:.3e}")
    assert worst < (TOLBF if dtype == BF16 else TOL32)
    gc = golden(f"chains_{preset}")
    L = int(gc["L"])
    inp = W.synth_inputs(preset, 1, L, seed=7)
    plan.prepare(inp["hub"], inp["style"], inp.get("emo"), L=L)
    noise = torch.from_numpy(gc["ddpm_lo_noise"])
    a = plan.sample_ddpm(inp["x"].to(DEV), gc["ddpm_lo_t"].tolist(), noise=noise, use_graph=True)
    b = plan.sample_ddpm(inp["x"].to(DEV), gc["ddpm_lo_t"].tolist(), noise=noise, use_graph=False)
    assert torch.equal(a, b)
    assert mad(a, gc["ddpm_lo_steps"][-1]) < (0.15 if dtype == BF16 else TOL32)
    B, L, t = 3, 37, 640
    inp = W.synth_inputs(preset, B, L, seed=21)
    plan.prepare(inp["hub"], inp["style"], inp.get("emo"), L=L)
    out = plan.denoise(inp["x"].to(DEV), t).cpu()
    plan.prepare(inp["hub"][2:3], inp["style"][2:3], None if "emo" not in inp else inp["emo"][2:3], L=L)
    assert torch.equal(plan.denoise(inp["x"][2:3].to(DEV), t).cpu()[0], out[2]), "per-clip result depends on the batch"
    plan.tune()          # the tuner's candidates at a K-sliced site are the 64-column tiles; tiles never change results
    assert torch.equal(plan.denoise(inp["x"][2:3].to(DEV), t).cpu()[0], out[2])
    with pytest.raises(Exception):
        plan.set("ksplit.out", 3)


@pytest.mark.parametrize("preset,dtype,cfg", [("mead", F16X3, True), ("vocaset", F16X3, False), ("mead", BF16, True), ("biwi", F32, False)])
def test_split_k_setting_composes_with_guidance_condition_batching_ddim_and_long_clips(preset, dtype, cfg):
    """The single-clip setting (K slices 2 / 4) under everything the step program composes with: classifier-free guidance (cond + uncond
    rows in one launch set), S conditions per clip (the LayerNorm launch that sums the planes also maps rows onto the clip's shared
    tables), the DDIM update, the longest clip (L = 600), head_dim 256 (BIWI).  Against the same plan without K slices: within fp32
    rounding in the parity modes (and against the oracle at 1e-4); condition blocks stay bit-identical to their own B = 1 calls."""
    w = W.make_fdm_weights(preset)
    plan = DenoiserPlan(preset, w, dtype, DEV)
    p = plan.p
    B, S, L, t = 1, 3, 41, 333
    inp = W.synth_inputs(preset, B, L, seed=31)
    hub = inp["hub"][:, :, :768].contiguous() if preset == "biwi" else inp["hub"]
    g = torch.Generator().manual_seed(4)
    style = torch.eye(p.n_style)[torch.randint(0, p.n_style, (B * S,), generator=g)]
    emo = torch.eye(p.n_emo)[torch.randint(0, p.n_emo, (B * S,), generator=g)] if p.n_emo else None
    x = torch.randn(B * S, L * p.G, p.c, generator=g)

    def run():
        plan.prepare(hub, style, emo, L=L, cfg=cfg, n_conds=S)
        return plan.denoise(x.to(DEV), t).cpu(), plan.sample_ddim(x.to(DEV), 6).cpu()
    d1, c1 = run()
    plan.set("ksplit.out", 2); plan.set("ksplit.ffn2", 4)
    d2, c2 = run()
    tol = 0.1 if dtype == BF16 else 2e-5
    assert mad(d1, d2) < tol and mad(c1, c2) < tol
    r = 2                                   # condition block 2 == its own B = 1 call, bit for bit, with the K slices on
    plan.prepare(hub, style[r:r + 1], None if emo is None else emo[r:r + 1], L=L, cfg=cfg)
    assert torch.equal(plan.denoise(x[r:r + 1].to(DEV), t).cpu()[0], d2[r])
    if preset != "biwi":                    # (BIWI's denoiser is build-defined: oracle-only elsewhere)
        ref = FO.fdm_forward_cfg(w, preset, hub, t, x[r:r + 1], style[r:r + 1], emo[r:r + 1], 2.5, folded=True) if cfg else \
            FO.fdm_forward(w, preset, hub, t, x[r:r + 1], style[r:r + 1], None if emo is None else emo[r:r + 1], folded=True)
        assert mad(d2[r:r + 1], ref) < (TOLBF if dtype == BF16 else TOL32)
    # the longest clip the model accepts
    L2 = 600
    inp2 = W.synth_inputs(preset, 1, L2, seed=32)
    hub2 = inp2["hub"][:, :, :768].contiguous() if preset == "biwi" else inp2["hub"]
    plan.prepare(hub2, inp2["style"], inp2.get("emo"), L=L2, cfg=cfg)
    a = plan.denoise(inp2["x"].to(DEV), 5).cpu()
    plan.set("ksplit.out", 1); plan.set("ksplit.ffn2", 1)
    plan.prepare(hub2, inp2["style"], inp2.get("emo"), L=L2, cfg=cfg)
    assert mad(a, plan.denoise(inp2["x"].to(DEV), 5).cpu()) < tol and torch.isfinite(a).all()


def test_ddpm_chain_with_device_noise_is_deterministic_and_shardable():
    """Philox noise is keyed by (seed, global clip index, step): a rank holding clips [1, 3) of a
    3-clip job reproduces those clips bit for bit."""
    plan, _ = plan_for("vocaset_tiny", F32)
    B, L = 3, 20
    inp = W.synth_inputs("vocaset_tiny", B, L, seed=33)
    ts = list(range(999, 979, -1))
    plan.prepare(inp["hub"], inp["style"], L=L)
    a = plan.sample_ddpm(inp["x"].to(DEV), ts, seed=77).cpu()
    a2 = plan.sample_ddpm(inp["x"].to(DEV), ts, seed=77).cpu()
    assert torch.equal(a, a2)
    plan.prepare(inp["hub"][1:], inp["style"][1:], L=L)
    s = plan.sample_ddpm(inp["x"][1:].to(DEV), ts, seed=77, clip0=1).cpu()
    assert torch.equal(s, a[1:])
    assert torch.isfinite(a).all()


def test_long_chain_bf16_stays_close_to_fp32():
    """Throughput mode: a 50-step DDIM chain in bf16 vs the fp32 plan (stated tolerance 0.15 max-abs)."""
    L = 24
    inp = W.synth_inputs("vocaset", 1, L, seed=5)
    outs = {}
    for dt in (F32, BF16):
        plan, _ = plan_for("vocaset", dt)
        plan.prepare(inp["hub"], inp["style"], L=L)
        outs[dt] = plan.sample_ddim(inp["x"].to(DEV), 50).cpu()
    assert mad(outs[F32], outs[BF16]) < 0.15


@pytest.mark.parametrize("dtype", PARITY_MODES)
def test_biwi_build_defined_semantics_vs_oracle(dtype):
    """BIWI denoiser (models/fdm.py is unrunnable as shipped): build-defined 'Dec' struct with the
    latent regrouped x8, style Mish, plain-Linear latent encoder, period-25 ALiBi.  Parity is pinned against the
    oracle restatement only (NOT against the reference).  head_dim 256: the split mode runs the K-then-V form of the split attention
    kernel (attention.hpp, attn_streamed; 58 launches per step like every other preset)."""
    preset = "biwi"
    w = W.make_fdm_weights(preset)
    plan = DenoiserPlan(preset, w, dtype, DEV)
    B, L, t = 2, 33, 412
    inp = W.synth_inputs(preset, B, L, seed=8)
    hub = torch.randn(B, 2 * L, 768, generator=torch.Generator().manual_seed(3))      # wav2vec2-base features
    plan.prepare(hub, inp["style"], L=L)
    out = plan.denoise(inp["x"].to(DEV), t).cpu()
    # oracle: pair the 768-wide frames -> 1536
    ref = []
    for b in range(B):
        a = hub[b].reshape(L, 1536)
        AFw = dict(w)
        ref.append(_biwi_oracle_clip(AFw, a, t, inp["x"][b], inp["style"][b]))
    assert mad(out, torch.stack(ref)) < TOL32


def _biwi_oracle_clip(w, a_paired, t, x, style):
    """Oracle forward for BIWI on already-paired audio rows [L, 1536]."""
    import torch.nn.functional as Fn
    p = W.PRESETS["biwi"]
    AF = Fn.linear(FO.mish(Fn.linear(a_paired, w["audio_extract.0.weight"], w["audio_extract.0.bias"])),
                   w["audio_extract.2.weight"], w["audio_extract.2.bias"])
    L = x.shape[0] // p["G"]
    h = Fn.linear(x.reshape(L, p["G"] * p["c"]), w["latent_encoder.0.weight"], w["latent_encoder.0.bias"])
    tau = FO.mish(w["time_embedd.0.weight"][:, t] + w["time_embedd.0.bias"])
    h = h + FO.mish(Fn.linear(style, w["style_embedd.weight"], w["style_embedd.bias"]))
    h = h + FO.positional_table(p["d"], "sinus", 25, L)
    mem = AF + tau
    mask = FO.biased_mask(p["n_head"], L, 25)
    d = p["d"]
    for l in range(p["n_layers"]):
        pre = f"transformer_decoder.layers.{l}."
        h = Fn.layer_norm(h + FO._mha_self(h, w, pre + "self_attn.", p["n_head"], mask), (d,), w[pre + "norm1.weight"], w[pre + "norm1.bias"], 1e-5)
        h = Fn.layer_norm(h + FO._mha_cross(h, mem, w, pre + "multihead_attn.", p["n_head"], True), (d,), w[pre + "norm2.weight"], w[pre + "norm2.bias"], 1e-5)
        f = Fn.linear(torch.relu(Fn.linear(h, w[pre + "linear1.weight"], w[pre + "linear1.bias"])), w[pre + "linear2.weight"], w[pre + "linear2.bias"])
        h = Fn.layer_norm(h + f, (d,), w[pre + "norm3.weight"], w[pre + "norm3.bias"], 1e-5)
    return Fn.linear(h, w["latent_decoder.weight"], w["latent_decoder.bias"]).reshape(L * p["G"], p["c"])


def test_bf16_folded_norm3_matches_unfolded():
    """bf16 step program with norm3 folded into the surrounding GEMMs (fdm_plan_set "fuse_ln3": opt-in, takes effect at the next
    commit) vs the same program with the LN3 launches."""
    L, t = 50, 777
    inp = W.synth_inputs("vocaset", 2, L, seed=31)
    w = W.make_fdm_weights("vocaset")
    outs = {}
    plan = DenoiserPlan("vocaset", w, BF16, DEV)
    assert plan.fuse_ln3 is False                                        # opt-in since the specialised GEMM kernels
    for flag in (0, 1, 0):
        plan.set("fuse_ln3", flag)
        plan.prepare(inp["hub"], inp["style"], L=L)
        assert plan.fuse_ln3 == bool(flag) and plan.get("launches_per_step") in (0, 58, 50)
        outs.setdefault(flag, []).append(plan.denoise(inp["x"].to(DEV), t).cpu())
    assert torch.equal(outs[0][0], outs[0][1])                            # switching back rebuilds the unfolded program exactly
    ref = FO.fdm_forward(w, "vocaset", inp["hub"], t, inp["x"], inp["style"], None, folded=True)
    assert mad(outs[1][0], ref) < TOLBF and mad(outs[0][0], ref) < TOLBF
    assert mad(outs[1][0], outs[0][0]) < TOLBF


def test_f16x3_folded_norm3_is_opt_in_and_stays_in_contract():
    """fdm_plan_set "fuse_ln3" folds norm3 into the surrounding GEMMs in the split-fp16 program too (the same algebra on plane
    pairs: 8 launches fewer -- off by default): still inside the 1e-4 contract of the reference goldens."""
    L, t = 50, 777
    inp = W.synth_inputs("vocaset", 2, L, seed=31)
    w = W.make_fdm_weights("vocaset")
    ref = FO.fdm_forward(w, "vocaset", inp["hub"], t, inp["x"], inp["style"], None, folded=True)
    for flag in (1, 0):
        plan = DenoiserPlan("vocaset", w, F16X3, DEV)
        assert plan.fuse_ln3 is False
        plan.set("fuse_ln3", flag)
        plan.prepare(inp["hub"], inp["style"], L=L)
        assert plan.fuse_ln3 == bool(flag)
        assert mad(plan.denoise(inp["x"].to(DEV), t).cpu(), ref) < TOL32


def test_full_size_cfg2_chain_properties():
    """BASELINE.json configs[1] at full size (4 clips x 200 frames, 1000 DDPM steps): the oracle cannot finish this in
    seconds, so the chain is checked through size-independent properties: run-to-run determinism, clip
    independence (a clip sampled alone == the same clip sampled in the batch, bit for bit, with its Philox stream
    keyed by the global clip index), captured graph == eager launches, and the bf16 program staying within its
    stated distance of the fp32 program after all 1000 steps."""
    B, L, T = 4, 200, 1000
    inp = W.synth_inputs("vocaset", B, L, seed=1)
    ts = list(range(T - 1, -1, -1))
    xT = inp["x"].to(DEV)
    outs = {}
    for dt in (F32, BF16, F16X3):
        plan, _ = plan_for("vocaset", dt)
        plan.prepare(inp["hub"], inp["style"], L=L)
        a = plan.sample_ddpm(xT, ts, seed=1234)
        b = plan.sample_ddpm(xT, ts, seed=1234)
        assert torch.equal(a, b), "not deterministic"
        assert torch.isfinite(a).all()
        outs[dt] = a.cpu()
        if dt == F32:
            e = plan.sample_ddpm(xT, ts[:40], seed=1234, use_graph=False)
            g2 = plan.sample_ddpm(xT, ts[:40], seed=1234, use_graph=True)
            assert torch.equal(e, g2), "graph replay differs from eager launches"
            plan.prepare(inp["hub"][2:3], inp["style"][2:3], L=L)
            one = plan.sample_ddpm(xT[2:3], ts, seed=1234, clip0=2).cpu()
            assert torch.equal(one[0], outs[F32][2]), "clip result depends on the batch it was sampled in"
    d = mad(outs[F32], outs[BF16])
    scale = float(outs[F32].abs().max())
    d3 = mad(outs[F32], outs[F16X3])
    print(f"full-size cfg2: max-abs after 1000 steps vs the fp32 program: bf16 {d:.3e}, f16x3 {d3:.3e} (latent max {scale:.2f})")
    assert d < 0.25
    assert d3 < TOL32, "the split-fp16 program left the contract tolerance over the full chain"


@pytest.mark.parametrize("dtype", [F32, BF16, F16X3])
def test_gemm_tile_choice_changes_speed_not_results(dtype, monkeypatch):
    """Plan-time tile tuning (fdm_plan_tune): every output tile accumulates k in the same order, so forcing any tile at
    every tuned call site gives bit-identical latents; FDM_TUNE=0 (library heuristic) likewise."""
    from fdm_amd._lib import TILE_64x64, TILE_64x64_S3, TILE_128x64_S3, TILE_96x128, TILE_64x64_S2, TILE_32x64_S3, TILE_128x64, TILE_128x128, TILE_256x128, TILE_256x128_PP, TILE_80x128, TILE_64x128
    from fdm_amd.denoiser import TILE_SITES
    L, t = 70, 432
    inp = W.synth_inputs("vocaset", 2, L, seed=5)
    plan, _ = plan_for("vocaset", dtype)
    plan.prepare(inp["hub"], inp["style"], L=L)
    plan.set("untune", 1)
    plan.tune()                        # forced (sampling calls tune lazily, once a shape has run 2000 steps)
    assert plan.get("tuned") == 1 and all(0 <= v <= 12 for v in plan.tiles.values())
    for k in TILE_SITES:
        plan.set("tile." + k, 0)
    base = plan.denoise(inp["x"].to(DEV), t).clone()
    for tile in (TILE_64x64, TILE_64x64_S3, TILE_64x64_S2, TILE_32x64_S3, TILE_128x64, TILE_128x64_S3, TILE_128x128, TILE_96x128, TILE_256x128, TILE_256x128_PP, TILE_80x128, TILE_64x128):
        for k in TILE_SITES:
            plan.set("tile." + k, tile)
        assert torch.equal(plan.denoise(inp["x"].to(DEV), t), base), f"tile {tile}"
    monkeypatch.setenv("FDM_TUNE", "0")
    plan.set("untune", 1)
    plan.prepare(inp["hub"], inp["style"], L=L)
    plan.tune()
    assert plan.get("tuned") == 0 and not any(plan.tiles.values())
    assert torch.equal(plan.denoise(inp["x"].to(DEV), t), base)
    monkeypatch.delenv("FDM_TUNE")
    plan.set("untune", 1)


@pytest.mark.parametrize("preset,L", [("vocaset", 498), ("mead", 300), ("vocaset", 600)])
def test_long_clip_single_step_vs_oracle(preset, L):
    """Clip lengths of the end-to-end configs (10 s audio -> L = 498: 32-query attention workgroups, >= 16 key tiles;
    cfg3's L = 300; the L = 600 limit of init_biased_mask) against the oracle, fp32 and bf16."""
    inp = W.synth_inputs(preset, 1, L, seed=900 + L)
    _, w = plan_for(preset, F32)
    t = 321
    ref = FO.fdm_forward(w, preset, inp["hub"], t, inp["x"], inp["style"], inp.get("emo"), folded=True)
    for dt, tol in ((F32, TOL32), (F16X3, TOL32), (BF16, TOLBF)):
        plan, _ = plan_for(preset, dt)
        plan.prepare(inp["hub"], inp["style"], inp.get("emo"), L=L)
        assert mad(plan.denoise(inp["x"].to(DEV), t).cpu(), ref) < tol, (preset, L, dt)


def test_full_length_mead_cfg_chain_properties():
    """cfg3's shape class (MEAD, L = 300, classifier-free guidance: cond + uncond rows in one set of launches) through
    size-independent properties: determinism, graph == eager, clip independence under CFG."""
    B, L = 2, 300
    inp = W.synth_inputs("mead", B, L, seed=3)
    ts = list(range(999, 879, -1))                # 120 steps
    plan, _ = plan_for("mead", F32)
    plan.prepare(inp["hub"], inp["style"], inp["emo"], L=L, cfg=True)
    plan.tune()                            # forced: the B = 2 and B = 1 plans below run with their own tuned tiles
    xT = inp["x"].to(DEV)
    a = plan.sample_ddpm(xT, ts, seed=9, cfg_scale=2.5)
    assert torch.equal(a, plan.sample_ddpm(xT, ts, seed=9, cfg_scale=2.5)) and torch.isfinite(a).all()
    assert torch.equal(plan.sample_ddpm(xT, ts[:30], seed=9, cfg_scale=2.5, use_graph=False),
                       plan.sample_ddpm(xT, ts[:30], seed=9, cfg_scale=2.5, use_graph=True))
    plan.prepare(inp["hub"][1:], inp["style"][1:], inp["emo"][1:], L=L, cfg=True)
    plan.tune()
    one = plan.sample_ddpm(xT[1:], ts, seed=9, clip0=1, cfg_scale=2.5)
    assert torch.equal(one[0], a[1]), "clip result depends on the batch it was sampled in"


# ---- condition-batched sampling: S conditions per clip in one step program (fdm_audio_prepare_conds) ----
@pytest.mark.parametrize("preset,dtype,cfg", [("vocaset", F32, False), ("vocaset", F16X3, False), ("vocaset", BF16, False),
                                              ("mead", F32, True), ("mead", F16X3, False)])
def test_condition_batched_equals_sequential_b1_calls(preset, dtype, cfg):
    """The reference's sampler runs the style one-hots of a clip as sequential B = 1 calls with the same audio
    (samples/sample_diffusion_vocaset.py:71-83).  One call with S conditions per clip: every (clip, condition) block is
    bit-identical to its own B = 1 call (denoiser pass, DDIM chain, DDPM chain with Philox keyed by clip0 = b*S + s), and
    within 1e-4 of the oracle on two conditions.  The audio tables are built once per CLIP (shared by its conditions)."""
    plan, w = plan_for(preset, dtype)
    p = plan.p
    B, S, L, t = 2, 4, 23, 417
    inp = W.synth_inputs(preset, B, L, seed=5)
    g = torch.Generator().manual_seed(9)
    style = torch.eye(p.n_style)[torch.randint(0, p.n_style, (B * S,), generator=g)]
    emo = torch.eye(p.n_emo)[torch.randint(0, p.n_emo, (B * S,), generator=g)] if p.n_emo else None
    x = torch.randn(B * S, L * p.G, p.c, generator=g)
    plan.prepare(inp["hub"], style, emo, L=L, cfg=cfg, n_conds=S)
    assert plan.B == B * S and plan.get("rows") == B * S * L * (2 if cfg else 1)
    out = plan.denoise(x.to(DEV), t).cpu()
    ddim = plan.sample_ddim(x.to(DEV), 5).cpu()
    ts = list(range(999, 989, -1))
    ddpm = plan.sample_ddpm(x.to(DEV), ts, seed=11, clip0=40).cpu()
    for (b, s) in ((0, 0), (0, 3), (1, 2)):
        r = b * S + s
        e1 = None if emo is None else emo[r:r + 1]
        plan.prepare(inp["hub"][b:b + 1], style[r:r + 1], e1, L=L, cfg=cfg)
        assert torch.equal(plan.denoise(x[r:r + 1].to(DEV), t).cpu()[0], out[r]), "denoiser pass differs from the B = 1 call"
        assert torch.equal(plan.sample_ddim(x[r:r + 1].to(DEV), 5).cpu()[0], ddim[r]), "DDIM chain differs from the B = 1 call"
        assert torch.equal(plan.sample_ddpm(x[r:r + 1].to(DEV), ts, seed=11, clip0=40 + r).cpu()[0], ddpm[r]), "DDPM chain differs"
    # vs the oracle (the reference's arithmetic) on two (clip, condition) blocks
    for (b, s) in ((0, 1), (1, 3)):
        r = b * S + s
        e1 = None if emo is None else emo[r:r + 1]
        ref = FO.fdm_forward(w, preset, inp["hub"][b:b + 1], t, x[r:r + 1], style[r:r + 1], e1, folded=True)
        if cfg:
            ref = FO.cfg_mix(ref, FO.fdm_forward(w, preset, inp["hub"][b:b + 1], t, x[r:r + 1], style[r:r + 1], torch.zeros_like(e1), folded=True))
        assert mad(out[r], ref[0]) < (TOLBF if dtype == BF16 else TOL32)


def test_condition_batching_validates_arguments():
    plan, _ = plan_for("vocaset_tiny", F32)
    inp = W.synth_inputs("vocaset_tiny", 2, 8, seed=1)
    from fdm_amd._lib import FdmError
    with pytest.raises(FdmError):
        plan.prepare(inp["hub"], inp["style"], L=8, n_conds=3)        # style rows != clips x conditions
    with pytest.raises(FdmError):
        plan.prepare(inp["hub"], inp["style"], L=8, n_conds=0)


def test_program_cache_never_evicts_a_program_in_use():
    """sweeping graph_steps overflows the program cache (16 entries, LRU, keyed by shape / tile set / kind); the
    1-step program fetched first must survive the insertion of the K-step program in the same call (it replays the
    n_steps % K tail)."""
    plan, _ = plan_for("vocaset_tiny", F32)
    L = 12
    inp = W.synth_inputs("vocaset_tiny", 1, L, seed=2)
    plan.prepare(inp["hub"], inp["style"], L=L)
    ts = list(range(999, 999 - 23, -1))
    ref = plan.sample_ddpm(inp["x"].to(DEV), ts, seed=3, use_graph=False).cpu()
    for k in list(range(2, 23)) + [2, 9, 22, 3]:
        out = plan.sample_ddpm(inp["x"].to(DEV), ts, seed=3, graph_steps=k).cpu()
        assert torch.equal(out, ref), f"graph_steps={k}"


def test_weight_update_rebuilds_tables_without_growing():
    """fdm_plan_set_weights after a commit releases the commit-time tables; results follow the new weights and
    device memory does not grow by a model footprint per update."""
    w = W.make_fdm_weights("vocaset_tiny")
    plan = DenoiserPlan("vocaset_tiny", w, F32, DEV)
    L = 10
    inp = W.synth_inputs("vocaset_tiny", 1, L, seed=4)
    plan.prepare(inp["hub"], inp["style"], L=L)
    a = plan.denoise(inp["x"].to(DEV), 500).cpu()
    import ctypes as C
    from fdm_amd._lib import check, lib
    w2 = {k: v.clone() for k, v in w.items()}
    w2["latent_decoder.weight"] = w2["latent_decoder.weight"] * 2.0
    torch.cuda.synchronize()
    base = None
    for it in range(6):
        src = w2 if it % 2 == 0 else w
        t = src["latent_decoder.weight"].contiguous()
        check(lib().fdm_plan_set_weights(plan.h, b"latent_decoder.weight", t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        plan.prepare(inp["hub"], inp["style"], L=L)
        out = plan.denoise(inp["x"].to(DEV), 500).cpu()
        ref = FO.fdm_forward(src, "vocaset_tiny", inp["hub"], 500, inp["x"], inp["style"], None, folded=True)
        assert mad(out, ref) < TOL32
        torch.cuda.synchronize()
        free, _ = torch.cuda.mem_get_info()
        if it == 1:
            base = free
        if it > 1:
            assert base - free < (1 << 20), f"device memory grew by {(base - free) >> 10} KiB after weight update {it}"
    assert mad(a, FO.fdm_forward(w, "vocaset_tiny", inp["hub"], 500, inp["x"], inp["style"], None, folded=True)) < TOL32


def test_tile_override_applies_at_prepare_and_tuning_is_plan_time(monkeypatch):
    """FDM_TILE_OVERRIDE pins call sites at fdm_audio_prepare whether or not the tuner runs (a C caller that only
    calls fdm_sample_graph gets the pinned tiles), and a request path (fdm_audio_prepare, fdm_sample_graph) never tunes by itself:
    tuning is fdm_plan_tune -- fdm_plan_get "needs_tune" says when a shape has served 2000 steps on heuristic tiles -- unless the
    caller opted in (fdm_plan_set(p, "tune_lazy", 1))."""
    w = W.make_fdm_weights("vocaset_tiny")
    L = 24
    inp = W.synth_inputs("vocaset_tiny", 2, L, seed=6)
    monkeypatch.setenv("FDM_TUNE", "0")
    monkeypatch.setenv("FDM_TILE_OVERRIDE", "out=2,ffn1=6")
    plan = DenoiserPlan("vocaset_tiny", w, F32, DEV)
    plan.prepare(inp["hub"], inp["style"], L=L)
    assert plan.tiles["out"] == 2 and plan.tiles["ffn1"] == 6 and plan.get("tuned") == 0
    ref = plan.sample_ddim(inp["x"].to(DEV), 20)
    monkeypatch.delenv("FDM_TILE_OVERRIDE")
    monkeypatch.delenv("FDM_TUNE")
    plan2 = DenoiserPlan("vocaset_tiny", w, F32, DEV)
    plan2.prepare(inp["hub"], inp["style"], L=L)
    ts = list(range(999, -1, -1))
    assert plan2.get("needs_tune") == 0
    for _ in range(3):                                  # 3000 steps at this shape: still untuned, sampling calls only count
        out = plan2.sample_ddpm(inp["x"].to(DEV), ts, seed=1)
        assert plan2.get("tuned") == 0
    assert plan2.get("needs_tune") == 1                 # ... and say so
    plan2.prepare(inp["hub"], inp["style"], L=L)       # a per-request call: no tuning latency spike here either
    assert plan2.get("tuned") == 0 and plan2.get("needs_tune") == 1
    plan2.tune()                                        # the caller schedules it off the request path
    assert plan2.get("tuned") == 1 and plan2.get("needs_tune") == 0 and plan2.get("tune_failed") == 0
    assert torch.equal(plan2.sample_ddpm(inp["x"].to(DEV), ts, seed=1), out)        # tiles change speed, never results
    assert torch.equal(plan2.sample_ddim(inp["x"].to(DEV), 20), ref)
    plan3 = DenoiserPlan("vocaset_tiny", w, F32, DEV)
    plan3.set("tune_lazy", 1)
    plan3.prepare(inp["hub"], inp["style"], L=L)
    for _ in range(3):
        plan3.sample_ddpm(inp["x"].to(DEV), ts, seed=1)
    assert plan3.get("tuned") == 1                      # opted in: the third call (2000 steps seen) tuned in-call


def test_tuned_tiles_persist_through_the_tile_cache_file(tmp_path, monkeypatch):
    """FDM_TILE_CACHE=<file> (opt-in): fdm_plan_tune writes the shape's tile set (keyed by library version, arithmetic mode, model
    geometry and shape); a NEW plan takes it at fdm_audio_prepare without tuning; another shape / mode / a damaged line does not
    match; the file never changes results."""
    path = tmp_path / "tiles.txt"
    monkeypatch.setenv("FDM_TILE_CACHE", str(path))
    w = W.make_fdm_weights("vocaset")
    L = 200
    inp = W.synth_inputs("vocaset", 4, L, seed=9)
    a = DenoiserPlan("vocaset", w, BF16, DEV)
    a.prepare(inp["hub"], inp["style"], L=L)
    assert a.get("tuned") == 0 and not path.exists()
    a.tune()
    assert a.get("tuned") == 1
    lines = path.read_text().splitlines()
    assert len(lines) == 1 and lines[0].startswith("v") and "\t" in lines[0]
    ref = a.denoise(inp["x"].to(DEV), 500).clone()
    b = DenoiserPlan("vocaset", w, BF16, DEV)
    b.prepare(inp["hub"], inp["style"], L=L)
    assert b.get("tuned") == 1 and b.tiles == a.tiles            # no fdm_plan_tune call on this plan
    assert torch.equal(b.denoise(inp["x"].to(DEV), 500), ref)
    b.prepare(inp["hub"][:2], inp["style"][:2], L=L)              # another shape: not in the file
    assert b.get("tuned") == 0
    c = DenoiserPlan("vocaset", w, F16X3, DEV)                    # another arithmetic mode: its own key
    c.prepare(inp["hub"], inp["style"], L=L)
    assert c.get("tuned") == 0
    c.tune()
    assert len(path.read_text().splitlines()) == 2
    key = lines[0].split("\t")[0]
    path.write_text(key + "\tqkv=99\n")                           # a damaged line is ignored (and replaced by the next tuning)
    d = DenoiserPlan("vocaset", w, BF16, DEV)
    d.prepare(inp["hub"], inp["style"], L=L)
    assert d.get("tuned") == 0
    assert torch.equal(d.denoise(inp["x"].to(DEV), 500), ref)
    monkeypatch.setenv("FDM_TILE_CACHE", str(tmp_path / "no_such_dir" / "tiles.txt"))      # unwritable: the store is off, the plan works
    e = DenoiserPlan("vocaset", w, BF16, DEV)
    e.prepare(inp["hub"], inp["style"], L=L)
    e.tune()
    assert e.get("tuned") == 1 and torch.equal(e.denoise(inp["x"].to(DEV), 500), ref)


def test_programs_are_kept_per_shape():
    """A serving loop that alternates between shapes finds its recorded programs again (keyed by shape and tile set) and gets the
    same bits as a fresh plan; a weight update still drops everything."""
    w = W.make_fdm_weights("vocaset_tiny")
    plan = DenoiserPlan("vocaset_tiny", w, F32, DEV)
    inpA, inpB = W.synth_inputs("vocaset_tiny", 2, 20, seed=1), W.synth_inputs("vocaset_tiny", 1, 33, seed=2)
    ts = list(range(999, 979, -1))
    refs = {}
    for rnd in range(3):
        for name, inp, L in (("A", inpA, 20), ("B", inpB, 33)):
            plan.prepare(inp["hub"], inp["style"], L=L)
            out = (plan.sample_ddpm(inp["x"].to(DEV), ts, seed=4).cpu(), plan.sample_ddim(inp["x"].to(DEV), 7).cpu())
            if name in refs:
                assert torch.equal(out[0], refs[name][0]) and torch.equal(out[1], refs[name][1]), (rnd, name)
            refs[name] = out
    fresh = DenoiserPlan("vocaset_tiny", w, F32, DEV)
    fresh.prepare(inpB["hub"], inpB["style"], L=33)
    assert torch.equal(fresh.sample_ddpm(inpB["x"].to(DEV), ts, seed=4).cpu(), refs["B"][0])
