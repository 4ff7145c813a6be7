"""GPU: BASELINE.json's configurations at their full per-GPU size, composed (the pieces are pinned one by one elsewhere).
The oracle cannot run these chains in seconds, so each is checked by (a) the first steps of the chain against the CPU oracle
on the same inputs and (b) size-independent properties over the FULL chain: run-to-run determinism, clip independence
(a clip sampled alone == the same clip sampled in the batch, bit for bit), finiteness, shapes.

  cfg3  3D-MEAD, 4 clips x 300 frames per GPU, 1000-step DDPM + classifier-free guidance (cond + uncond rows in one launch set)
  cfg4  BIWI, 4 clips x 200 frames per GPU, 250-step DDIM (249 live denoiser calls), head_dim 256 -- the BIWI denoiser is
        build-defined (models/fdm.py cannot run as shipped): pinned against the oracle restatement only, parity UNPINNED vs
        the reference; its DDIM schedule and update are pinned (tests/test_plan_host_cpu.py, chains goldens)
  cfg5  VOCASET end to end: 10 s of audio -> HuBERT-large -> per-clip tables -> 1000-step DDPM at L = 498 -> VQ quant ->
        decode to [4, 498, 15069]"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from fdm_amd._lib import BF16, F16, F16X3, F32  # noqa: E402
from fdm_amd.denoiser import DenoiserPlan  # noqa: E402
from oracle import fdm_oracle as FO  # noqa: E402
from oracle import weights as W  # noqa: E402

DEV = "cuda:0"
TOL = 1e-4
# bf16 (BASELINE configs name it; outside the 1e-4 contract by construction): stated bar = 2x the distance measured on MI355X
# (round 4, printed by the tests below with -s: 9.2e-5 / 1.44e-4 / 2.56e-4) over the first steps of each chain -- at t = 999..
# the update coefficients are tiny
BF16_FIRST_STEPS_BAR = {"cfg2": 2e-4, "cfg3": 3e-4, "cfg4": 5.2e-4}


# UN-ATTENUATED checks at the benched shapes (round 5).  The first steps of a DDPM chain multiply the denoiser's x0_hat by
# posterior_mean_coef1[999] ~ 1.6e-3 (video_diffusion_pytorch/diffusion_BIWI_encoder_decoder.py:632-639), so the bars above hold
# an attenuated quantity; the LAST steps (t = 2, 1, 0: coef1 = 0.45, 0.69, 1.0) and a direct denoiser call do not.  bf16 bars =
# 2x the distance measured on MI355X (printed by the tests with -s); f32 / f16x3 state the contract's 1e-4.
# measured (round 5): 2.39e-2 / 2.33e-2 / 5.00e-2 / 2.39e-2 / 2.48e-2; f32 1.0e-5 / 9.3e-6 / 1.6e-5 / 8.1e-6 / 7.2e-6; f16x3 6.0e-6 / 5.9e-6 / 9.8e-6 / 6.8e-6 / 5.0e-6
BF16_UNATTENUATED_BAR = {"cfg2_last": 4.8e-2, "cfg2_denoise": 4.7e-2, "cfg3_last": 0.1, "cfg4_last": 4.8e-2, "cfg5_last": 5.0e-2}


# single-plane fp16 (FDM_F16, round 6), same places, 2x measured (3.11e-3 / 3.07e-3; tools/measure_f16_bars.py, profiles/r6_f16/bars.txt)
F16_UNATTENUATED_BAR = {"cfg2_last": 6.2e-3, "cfg2_denoise": 6.2e-3}


def _ubar(dtype, key):
    return BF16_UNATTENUATED_BAR[key] if dtype == BF16 else (F16_UNATTENUATED_BAR[key] if dtype == F16 else TOL)


def mad(a, b):
    return float((torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max())


def _bar(dtype, cfg):
    return BF16_FIRST_STEPS_BAR[cfg] if dtype == BF16 else TOL


@pytest.mark.parametrize("dtype", [F32, BF16, F16X3])
def test_cfg2_first_steps_vs_oracle(dtype):
    """cfg2 at its benched shape (VOCASET, 4 clips x 200 frames, DDPM): the first 3 steps (t = 999, 998, 997; injected noise) of
    clips 0 and 3 against the CPU oracle, in every mode bench.py times."""
    preset, B, L, T = "vocaset", 4, 200, 1000
    w = W.make_fdm_weights(preset)
    inp = W.synth_inputs(preset, B, L, seed=2)
    plan = DenoiserPlan(preset, w, dtype, DEV)
    plan.prepare(inp["hub"], inp["style"], L=L)
    ts = list(range(T - 1, -1, -1))
    k = 3
    noise = torch.randn(k, *inp["x"].shape, generator=torch.Generator().manual_seed(0))
    rec = []
    plan.sample_ddpm(inp["x"].to(DEV), ts[:k], noise=noise, record=rec)
    worst = 0.0
    for b in (0, 3):
        den = lambda x, t: FO.fdm_forward(w, preset, inp["hub"][b:b + 1], t, x, inp["style"][b:b + 1], None, folded=True)
        ref = []
        FO.p_sample_loop(den, inp["x"][b:b + 1].clone(), noise[:, b:b + 1], ts[:k], record=ref)
        worst = max(worst, mad(torch.stack(rec)[:, b:b + 1], torch.stack(ref)))
    print(f"[cfg2 dtype {dtype}] first {k} steps vs oracle: max-abs {worst:.3e} (bar {_bar(dtype, 'cfg2'):.1e})")
    assert worst < _bar(dtype, "cfg2")


@pytest.mark.parametrize("dtype", [F32, BF16, F16X3, F16])
def test_cfg2_last_steps_and_direct_denoise_vs_oracle(dtype):
    """cfg2's program at its benched shape (4 clips x 200 frames), un-attenuated: the LAST three DDPM steps (t = 2, 1, 0, injected
    noise, seeded x) and one direct denoiser call at t = 500, clips 0 and 3, against the CPU oracle
    (video_diffusion_pytorch/diffusion_BIWI_encoder_decoder.py:632-656; models/fdm_vocaset.py:56-93)."""
    preset, B, L = "vocaset", 4, 200
    w = W.make_fdm_weights(preset)
    inp = W.synth_inputs(preset, B, L, seed=2)
    plan = DenoiserPlan(preset, w, dtype, DEV)
    plan.prepare(inp["hub"], inp["style"], L=L)
    ts = [2, 1, 0]
    noise = torch.randn(3, *inp["x"].shape, generator=torch.Generator().manual_seed(0))
    rec = []
    plan.sample_ddpm(inp["x"].to(DEV), ts, noise=noise, record=rec)
    x0 = plan.denoise(inp["x"].to(DEV), 500)
    worst = worst_d = 0.0
    for b in (0, 3):
        den = lambda x, t: FO.fdm_forward(w, preset, inp["hub"][b:b + 1], t, x, inp["style"][b:b + 1], None, folded=True)
        ref = []
        FO.p_sample_loop(den, inp["x"][b:b + 1].clone(), noise[:, b:b + 1], ts, record=ref)
        worst = max(worst, mad(torch.stack(rec)[:, b:b + 1], torch.stack(ref)))
        worst_d = max(worst_d, mad(x0[b:b + 1], den(inp["x"][b:b + 1], 500)))
    print(f"[cfg2 dtype {dtype}] last 3 steps (t = 2, 1, 0) vs oracle: max-abs {worst:.3e} (bar {_ubar(dtype, 'cfg2_last'):.1e}); "
          f"denoise(t = 500) {worst_d:.3e} (bar {_ubar(dtype, 'cfg2_denoise'):.1e})")
    assert worst < _ubar(dtype, "cfg2_last") and worst_d < _ubar(dtype, "cfg2_denoise")


@pytest.mark.parametrize("dtype", [F32, BF16, F16X3])
def test_cfg3_mead_full_chain_with_guidance(dtype):
    preset, B, L, T = "mead", 4, 300, 1000
    w = W.make_fdm_weights(preset)
    inp = W.synth_inputs(preset, B, L, seed=3)
    plan = DenoiserPlan(preset, w, dtype, DEV)
    plan.prepare(inp["hub"], inp["style"], inp["emo"], L=L, cfg=True)
    xT = inp["x"].to(DEV)
    ts = list(range(T - 1, -1, -1))
    # first steps vs the oracle's two-pass composition (injected noise), clip 1 only (the oracle runs ~1 s per forward here)
    k = 3
    noise = torch.randn(k, *inp["x"].shape, generator=torch.Generator().manual_seed(0))
    rec = []
    plan.sample_ddpm(xT, ts[:k], noise=noise, cfg_scale=2.5, record=rec)
    den = lambda x, t: FO.fdm_forward_cfg(w, preset, inp["hub"][1:2], t, x, inp["style"][1:2], inp["emo"][1:2], 2.5, folded=True)
    ref = []
    FO.p_sample_loop(den, inp["x"][1:2].clone(), noise[:, 1:2], ts[:k], record=ref)
    dist = mad(torch.stack(rec)[:, 1:2], torch.stack(ref))
    print(f"[cfg3 dtype {dtype}] first {k} steps vs oracle: max-abs {dist:.3e} (bar {_bar(dtype, 'cfg3'):.1e})")
    assert dist < _bar(dtype, "cfg3")
    # un-attenuated: the last three steps (t = 2, 1, 0) of the guided chain from the seeded x, same clip
    rec = []
    plan.sample_ddpm(xT, ts[-k:], noise=noise, cfg_scale=2.5, record=rec)
    ref = []
    FO.p_sample_loop(den, inp["x"][1:2].clone(), noise[:, 1:2], ts[-k:], record=ref)
    dist = mad(torch.stack(rec)[:, 1:2], torch.stack(ref))
    print(f"[cfg3 dtype {dtype}] last {k} steps (t = 2, 1, 0) vs oracle: max-abs {dist:.3e} (bar {_ubar(dtype, 'cfg3_last'):.1e})")
    assert dist < _ubar(dtype, "cfg3_last")
    # full chain: determinism, finiteness, clip independence under CFG
    a = plan.sample_ddpm(xT, ts, seed=9, cfg_scale=2.5)
    assert a.shape == (B, L * 8, 64) and torch.isfinite(a).all()
    assert torch.equal(a, plan.sample_ddpm(xT, ts, seed=9, cfg_scale=2.5)), "not deterministic"
    plan.prepare(inp["hub"][2:3], inp["style"][2:3], inp["emo"][2:3], L=L, cfg=True)
    one = plan.sample_ddpm(xT[2:3], ts, seed=9, clip0=2, cfg_scale=2.5)
    assert torch.equal(one[0], a[2]), "clip result depends on the batch it was sampled in"


@pytest.mark.parametrize("dtype", [F32, BF16, F16X3])
def test_cfg4_biwi_full_ddim_chain(dtype):
    from test_denoiser_gpu import _biwi_oracle_clip
    preset, B, L, steps = "biwi", 4, 200, 250
    w = W.make_fdm_weights(preset)
    inp = W.synth_inputs(preset, B, L, seed=8)
    hub = torch.randn(B, 2 * L, 768, generator=torch.Generator().manual_seed(3))      # wav2vec2-base features
    plan = DenoiserPlan(preset, w, dtype, DEV)
    assert plan.p.head_dim == 256
    plan.prepare(hub, inp["style"], L=L)
    xT = inp["x"].to(DEV)
    rec = []
    a = plan.sample_ddim(xT, steps, record=rec)
    pairs = [pr for pr in FO.ddim_time_pairs(steps) if pr[1] >= 0]
    assert len(rec) == len(pairs) == 249 and torch.equal(rec[-1], a)
    assert a.shape == (B, L * 8, 128) and torch.isfinite(a).all()
    # first two live pairs vs the oracle (build-defined BIWI semantics), clip 3
    buf = FO.schedule_buffers()
    x = inp["x"][3].clone()
    for i, (t, tn) in enumerate(pairs[:2]):
        x0 = _biwi_oracle_clip(w, hub[3].reshape(L, 1536), t, x, inp["style"][3])
        x = FO.ddim_step(buf, x0, x, t, tn)
        dist = mad(rec[i][3], x)
        print(f"[cfg4 dtype {dtype}] live pair {i} (t = {t}) vs oracle: max-abs {dist:.3e} (bar {_bar(dtype, 'cfg4'):.1e})")
        assert dist < _bar(dtype, "cfg4"), (i, t)
    # un-attenuated: the last two LIVE pairs of the schedule (t = 8 -> 4 -> 0) from the seeded x, clip 3.  One denoiser call each
    # through the plan, the oracle's DDIM update applied to both sides' x0_hat
    x = inp["x"][3].clone()
    xs = inp["x"].clone()
    for (t, tn) in pairs[-2:]:
        x0h = plan.denoise(xs.to(DEV), t).cpu()
        x0o = _biwi_oracle_clip(w, hub[3].reshape(L, 1536), t, x, inp["style"][3])
        dist = mad(x0h[3], x0o)
        print(f"[cfg4 dtype {dtype}] last live pairs, denoise(t = {t}) vs oracle: max-abs {dist:.3e} (bar {_ubar(dtype, 'cfg4_last'):.1e})")
        assert dist < _ubar(dtype, "cfg4_last"), t
        x = FO.ddim_step(buf, x0o, x, t, tn)
        xs[3] = x
    # properties over the whole 249-call chain
    assert torch.equal(a, plan.sample_ddim(xT, steps)), "not deterministic"
    assert torch.equal(a, plan.sample_ddim(xT, steps, graph_steps=1)), "steps per graph launch changed the result"
    plan.prepare(hub[1:2], inp["style"][1:2], L=L)
    assert torch.equal(plan.sample_ddim(xT[1:2], steps)[0], a[1]), "clip result depends on the batch it was sampled in"


# cfg5 in every arithmetic mode bench.py offers for it.  (denoiser mode, once-per-clip stages' mode, bar on the first 5 steps of
# the chain vs the oracle fed the SAME HIP audio features).  The bf16 bar is 2x the measured distance (1.5e-4 over these 5 steps at t = 999..995, whose update coefficients are tiny; round 3);
# the parity modes state the contract's 1e-4.  f16x3 runs HuBERT with split-fp16 layers and quant / decode in fp32 (bench.py and the drop-in modules do the same); "f16x3_all"
# adds the VQ decoder's transformer on split-fp16 operands (VQPlan's own f16x3 mode).
CFG5_MODES = {"f32": (F32, F32, F32, TOL), "bf16": (BF16, BF16, BF16, 3e-4), "f16x3": (F16X3, F16X3, F32, TOL), "f16x3_all": (F16X3, F16X3, F16X3, TOL)}


@pytest.mark.parametrize("mode", ["f32", "bf16", "f16x3"])      # ("f16x3_all" -- the VQ decoder on split operands too -- is pinned by test_vq_in_the_contract_mode_vs_golden)
def test_cfg5_vocaset_end_to_end_composed(mode):
    from fdm_amd.hubert import HubertPlan
    from fdm_amd.vq import VQPlan
    dt, hub_dt, side_dt, bar = CFG5_MODES[mode]
    preset, B, T = "vocaset", 4, 1000
    w = W.make_fdm_weights(preset)
    wav = (torch.randn(B, 160000, generator=torch.Generator().manual_seed(100)) * 0.1).to(DEV)
    hub_plan = HubertPlan(W.make_hubert_weights(24), 24, hub_dt, DEV)
    vq_plan = VQPlan(preset, W.make_vq_weights(preset), side_dt, DEV)
    hub = hub_plan.forward(wav)
    assert hub.shape == (B, 498, 1024) and torch.isfinite(hub).all()
    L = 498
    inp = W.synth_inputs(preset, B, L, seed=1)
    plan = DenoiserPlan(preset, w, dt, DEV)
    plan.prepare(hub, inp["style"], L=L)
    xT = inp["x"].to(DEV)
    ts = list(range(T - 1, -1, -1))
    # first 5 steps vs the oracle (fed the HIP HuBERT features; HuBERT itself is pinned by tests/golden/hubert.npz), clips 0 and 3
    k = 5
    noise = torch.randn(k, *inp["x"].shape, generator=torch.Generator().manual_seed(0))
    rec = []
    plan.sample_ddpm(xT, ts[:k], noise=noise, record=rec)
    hub_c = hub.cpu()
    worst = 0.0
    for b in (0, 3):
        den = lambda x, t: FO.fdm_forward(w, preset, hub_c[b:b + 1], t, x, inp["style"][b:b + 1], None, folded=True)
        ref = []
        FO.p_sample_loop(den, inp["x"][b:b + 1].clone(), noise[:, b:b + 1], ts[:k], record=ref)
        worst = max(worst, mad(torch.stack(rec)[:, b:b + 1], torch.stack(ref)))
    print(f"[cfg5 {mode}] first {k} steps vs oracle: max-abs {worst:.3e} (bar {bar:.1e})")
    assert worst < bar, mode
    # un-attenuated: the last three steps (t = 2, 1, 0) at L = 498 from the seeded x, clip 0
    rec = []
    plan.sample_ddpm(xT, ts[-3:], noise=noise[:3], record=rec)
    den = lambda x, t: FO.fdm_forward(w, preset, hub_c[0:1], t, x, inp["style"][0:1], None, folded=True)
    ref = []
    FO.p_sample_loop(den, inp["x"][0:1].clone(), noise[:3, 0:1], ts[-3:], record=ref)
    last = mad(torch.stack(rec)[:, 0:1], torch.stack(ref))
    ubar = _ubar(dt, "cfg5_last")
    print(f"[cfg5 {mode}] last 3 steps (t = 2, 1, 0) vs oracle: max-abs {last:.3e} (bar {ubar:.1e})")
    assert last < ubar, mode

    def run(clips, clip0):
        plan.prepare(hub[clips], inp["style"][clips], L=L)
        lat = plan.sample_ddpm(xT[clips], ts, seed=1234, clip0=clip0)
        zq, idx = vq_plan.quant(lat * (1.5 / 1024))
        return lat, idx, vq_plan.decode(zq)
    lat, idx, verts = run(slice(0, B), 0)
    assert lat.shape == (B, L * 16, 64) and verts.shape == (B, L, 15069) and idx.numel() == B * L * 16
    assert torch.isfinite(lat).all() and torch.isfinite(verts).all() and int(idx.min()) >= 0 and int(idx.max()) < 256
    lat2, idx2, verts2 = run(slice(0, B), 0)
    assert torch.equal(lat, lat2) and torch.equal(idx, idx2) and torch.equal(verts, verts2), "not deterministic"
    lat1, idx1, verts1 = run(slice(2, 3), 2)
    assert torch.equal(lat1[0], lat[2]) and torch.equal(verts1[0], verts[2]), "clip result depends on the batch it was sampled in"
    assert torch.equal(idx1.reshape(-1), idx.reshape(B, -1)[2])
