"""GPU: the drop-in class surface end to end.  cfg-1 (BASELINE.json configs[0]): 1 clip x 100 frames,
DDIM 50 steps, HuBERT-large -- the reference ran it AS WRITTEN (HuBERT inside the loop, full
cross-attention; tests/golden/cfg1_e2e.npz); the HIP path runs hoisted + folded + dead-call-skipped
and must agree within 1e-4 max-abs."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "face-diffusion-model_amd", "dropin"))

from oracle import hubert_oracle as HO  # noqa: E402
from oracle import vq_oracle as VO  # noqa: E402
from oracle import weights as W  # noqa: E402

DEV = "cuda:0"


def mad(a, b):
    return float((torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max())


def test_cfg1_end_to_end_through_dropin_classes(golden):
    from models.fdm_vocaset import FDM
    from models.utils.config import vocaset_vq_vae_args
    from models.vq_vae_vocaset import VQAutoEncoder
    from video_diffusion_pytorch.diffusion_BIWI_encoder_decoder import GaussianDiffusion
    g = golden("cfg1_e2e")
    model = FDM(feature_dim=1024)
    diffusion = GaussianDiffusion(model, timesteps=1000, loss_type="l2")
    sd = {"denoise_fn." + k: v for k, v in W.make_fdm_weights("vocaset").items()}
    sd.update({"denoise_fn.audio_encoder." + k: v for k, v in W.make_hubert_weights(24).items()})
    res = diffusion.load_state_dict(sd, strict=False)           # checkpoint layout of the reference ('model' dict)
    assert not res.unexpected_keys
    gen = torch.Generator().manual_seed(1)
    wav = HO.processor_normalize(torch.randn(32080, generator=gen) * 0.1).unsqueeze(0).to(DEV)
    xT = torch.randn(1, 1600, 64, generator=gen)
    sid = torch.eye(8)[2:3].to(DEV)
    num_frames = model.audio_encoder(wav, "vocaset").last_hidden_state.shape[1]     # samples/sample_diffusion_vocaset.py:76
    assert num_frames == 100
    out = diffusion.ddim_sample(wav, (1, num_frames * 16, 64), sid, 50, x_T=xT)
    assert mad(out, g["final"]) < 1e-4
    # FDM.forward (one call, reference signature) == first denoiser call of that chain
    x0 = model(wav, torch.full((1,), 999, dtype=torch.long, device=DEV), xT.to(DEV), sid)
    assert x0.shape == (1, 1600, 64) and torch.isfinite(x0).all()
    # quant + decode on the result, against the oracle
    ae = VQAutoEncoder(vocaset_vq_vae_args())
    wv = W.make_vq_weights("vocaset")
    ae.load_state_dict(wv, strict=False)
    lat = out * (1.5 / 256 / 4)                                  # bring latents to the codebook scale
    quanted, _, info = ae.quant(lat)
    ozq, oidx = VO.quant(wv, "vocaset", lat.cpu())
    assert torch.equal(info[2].cpu(), oidx)
    verts = ae.decode(quanted)
    assert verts.shape == (1, 100, 15069)
    assert mad(verts, VO.decode(wv, "vocaset", ozq)) < 1e-4


def test_mead_sample_with_cfg_wrapper_and_p_sample():
    from models.fdm_vqvae_mead import FDM
    from utiles.classifierfree import ClassifierFreeSampleModel
    from video_diffusion_pytorch.diffusion_mead_encoder_decoder import GaussianDiffusion
    from oracle import fdm_oracle as FO
    model = FDM(feature_dim=512, audio_encoder=False)
    w = W.make_fdm_weights("mead")
    model.load_state_dict(w, strict=False)
    L = 10
    inp = W.synth_inputs("mead", 1, L, seed=3)
    model.set_audio_features(inp["hub"].to(DEV))
    audio = torch.zeros(1, 16, device=DEV)
    diff = GaussianDiffusion(model, timesteps=1000, loss_type="l2")
    emo, sid = inp["emo"].to(DEV), inp["style"].to(DEV)
    z = torch.randn(1, L * 8, 64, generator=torch.Generator().manual_seed(0))
    # p_sample with injected noise == oracle ddpm step
    x1 = diff.p_sample(inp["x"].to(DEV), torch.full((1,), 700, dtype=torch.long, device=DEV), audio, emo, sid, noise=z.to(DEV))
    buf = FO.schedule_buffers()
    x0 = FO.fdm_forward(w, "mead", inp["hub"], 700, inp["x"], inp["style"], inp["emo"], folded=True)
    assert mad(x1, FO.ddpm_step(buf, x0, inp["x"], 700, z)) < 1e-4
    # short chain through .sample(t_range=...) with the CFG wrapper (two passes batched, mix fused in the scheduler)
    cfgd = GaussianDiffusion(ClassifierFreeSampleModel(model, 2.5), timesteps=1000, loss_type="l2")
    ts = list(range(999, 994, -1))
    noise = torch.randn(len(ts), 1, L * 8, 64, generator=torch.Generator().manual_seed(1))
    out = cfgd.sample(audio, (1, L * 8, 64), emo, sid, noise=(inp["x"], noise), t_range=(999, 994))
    den = lambda x, t: FO.fdm_forward_cfg(w, "mead", inp["hub"], t, x, inp["style"], inp["emo"], 2.5, folded=True)
    ref = FO.p_sample_loop(den, inp["x"].clone(), noise, ts, buf)
    assert mad(out, ref) < 1e-4


def test_demo_cli_writes_reference_output_layout(tmp_path):
    """demo_vocaset.py flags + np.save(<audio_path>/<stem>.npy, [1, L, V3]) on a synthetic wav (DDIM 3 steps)."""
    from scipy.io import wavfile
    from fdm_amd import pipeline
    wav = (np.random.default_rng(0).standard_normal(16000) * 3000).astype(np.int16)
    wp = str(tmp_path / "hello.wav")
    wavfile.write(wp, 16000, wav)
    dst = pipeline.demo_main("vocaset", ["--audio_file", wp, "--audio_path", str(tmp_path / "result"), "--ddim_steps", "3"])
    arr = np.load(dst)
    # 1 s audio + 1 s zero pad = 32000 samples -> 98 HuBERT frames -> 98 latent frames
    assert arr.shape == (1, 98, 15069) and arr.dtype == np.float32 and np.isfinite(arr).all()


def test_style_loop_batched_equals_the_reference_loop(tmp_path):
    """The sampler of samples/sample_diffusion_vocaset.py:59-88 with every style one-hot of a clip: ONE condition-batched call
    per clip (audio encoder once, audio tables once, 8 conditions in one step program) writes the same files with the same
    bits as the reference's loop of eight B = 1 calls (audio -> HuBERT -> ddim_sample -> quant -> decode + template)."""
    import importlib.util
    import os
    import sys
    from fdm_amd import pipeline, presets
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sd = os.path.join(here, "face-diffusion-model_amd", "dropin", "samples")
    sys.path.insert(0, os.path.dirname(sd))
    spec = importlib.util.spec_from_file_location("sample_diffusion", os.path.join(sd, "sample_diffusion.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    p = presets.get("vocaset")
    diffusion, ae = pipeline.build_models("vocaset", None, DEV)
    enc = diffusion.denoise_fn.audio_encoder
    calls = {"n": 0}
    orig = enc.forward

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    enc.forward = counting
    # a VOCASET test-set file name: <subject>_<sentence>.wav
    subject, sentence = "FaceTalk_170809_00138_TA", "sentence21"
    for mode, batched in (("batched", True), ("sequential", False)):
        calls["n"] = 0
        mod.sample_step(mod.synthetic_loader(p, 1, 1.0, names=[f"{subject}_{sentence}.wav"]), DEV, diffusion, ae, str(tmp_path / mode), p, 4,
                        all_styles=True, batched=batched, dataset="vocaset")
        # batched: the audio encoder runs once per CLIP; the loop encodes once per call (the reference even re-encodes per step)
        assert calls["n"] == (1 if batched else p.n_style), f"{mode}: the audio encoder ran {calls['n']} times for one clip"
    # the reference's file names: <file>_condition_<conditioning training subject> (samples/sample_diffusion_vocaset.py:61-62,86-88)
    conds = mod.CONDITION_SUBJECTS["vocaset"]
    assert len(conds) == p.n_style and conds[0] == "FaceTalk_170728_03272_TA" and conds[7] == "FaceTalk_170912_03278_TA"
    for it in range(p.n_style):
        a = np.load(str(tmp_path / "batched" / f"{subject}_{sentence}_condition_{conds[it]}.npy"))
        b = np.load(str(tmp_path / "sequential" / f"{subject}_{sentence}_condition_{conds[it]}.npy"))
        assert a.shape == b.shape == (1, 48, 15069) and np.array_equal(a, b), it
    a0 = np.load(str(tmp_path / "batched" / f"{subject}_{sentence}_condition_{conds[0]}.npy"))
    a5 = np.load(str(tmp_path / "batched" / f"{subject}_{sentence}_condition_{conds[5]}.npy"))
    assert not np.array_equal(a0, a5)          # the style does change the animation
    # ... which are the names the evaluation drop-in reads back (computer_metrix.py:171-174): the diversity metric runs on the
    # sampler's output directory as it is
    from fdm_amd import metrics
    div = metrics.diversity(str(tmp_path / "batched"), " ".join(conds), subject, dataset="vocaset", device=DEV, verbose=False,
                            sentences=[sentence], nr_vertices=5023)
    assert np.isfinite(div) and div > 0


def test_clips_of_different_lengths_batch_exactly():
    """pipeline.animate_many: a test set's clips of different durations in ONE sampling call (padded at the end; the denoiser's
    attention is causal, everything else per row) -- each clip's latent and vertices are bit-identical to animating it alone
    (DDIM, the VOCASET sampler's mode), and for DDPM to the B = 1 call that draws the clip's own noise stream."""
    from fdm_amd import pipeline, presets
    p = presets.get("vocaset")
    diffusion, ae = pipeline.build_models("vocaset", None, DEV)
    g = torch.Generator().manual_seed(3)
    audios = [pipeline.processor_normalize((torch.randn(n, generator=g) * 0.1).numpy(), pad_seconds=0) for n in (16000, 24400, 11300, 20000)]
    ids = [torch.eye(p.n_style)[i:i + 1] for i in (2, 0, 7, 5)]
    tmpl = [torch.full((1, p.V3), 0.01 * i) for i in range(4)]
    verts, lats = pipeline.animate_many(diffusion, ae, audios, tmpl, ids, ddim_steps=6, device=DEV, max_batch=3)
    assert [v.shape[1] for v in verts] == [48, 76, 34, 62]
    for b in range(4):
        v1, l1 = pipeline.animate(diffusion, ae, audios[b], tmpl[b], ids[b], ddim_steps=6, device=DEV)
        assert torch.equal(l1, lats[b]) and torch.equal(v1, verts[b]), b
    # DDPM (short chain through t_range is not exposed by animate: use the module surface): clip b draws noise stream b
    model = diffusion.denoise_fn
    hubs = [model.audio_encoder(torch.as_tensor(a, device=DEV).reshape(1, -1)).last_hidden_state for a in audios[:3]]
    Ls = [h.shape[1] for h in hubs]
    Lmax = max(Ls)
    hub = torch.zeros(3, Lmax, 1024, device=DEV)
    xT = torch.zeros(3, Lmax * p.G, p.c)
    for i in range(3):
        hub[i, :Ls[i]] = hubs[i][0]
        xT[i, :Ls[i] * p.G] = torch.randn(Ls[i] * p.G, p.c, generator=torch.Generator().manual_seed(40 + i))
    model.set_audio_features(hub)
    both = diffusion.p_sample_loop((3, Lmax * p.G, p.c), torch.zeros(3, 1, device=DEV), torch.cat(ids[:3]).to(DEV), seed=9, x_T=xT, t_range=(999, 979))
    for i in range(3):
        model.set_audio_features(hubs[i])
        one = diffusion.p_sample_loop((1, Ls[i] * p.G, p.c), torch.zeros(1, 1, device=DEV), ids[i].to(DEV), seed=9, x_T=xT[i:i + 1, :Ls[i] * p.G],
                                      t_range=(999, 979), clip0=i)
        assert torch.equal(one[0], both[i, :Ls[i] * p.G]), i
    model._hub_key, model._hub = None, None


def test_mead_end_to_end_animate_with_evq():
    """3D-MEAD wiring of samples/sample_diffusion_mead.py:67-86: sample(audio, shape, emo, id) -> quant(result, emo)
    -> decode; short chain (t_range) on random-init weights; checks shapes, finiteness, emotion-sliced codebook use
    and that the quantised latents decode like the oracle."""
    from fdm_amd import pipeline
    diffusion, ae = pipeline.build_models("mead", device=DEV)
    model = diffusion.denoise_fn
    gen = torch.Generator().manual_seed(5)
    wav = HO.processor_normalize(torch.randn(2, 16000, generator=gen)[0] * 0.1)
    wav = torch.stack([wav, wav.flip(0)]).to(DEV)
    hub = model.audio_features(wav)
    L = hub.shape[1] // 2
    assert L == 24
    emo = torch.eye(7)[[2, 5]].to(DEV)
    sid = torch.eye(25)[[0, 3]].to(DEV)
    latent = diffusion.sample(wav, (2, L * 8, 64), emo, sid, seed=3, t_range=(999, 989))
    assert latent.shape == (2, L * 8, 64) and torch.isfinite(latent).all()
    lat = latent * (1.5 / 256 / 4)
    quanted, _, info = ae.quant(lat, emo)
    wv = {k: v.detach().cpu() for k, v in ae.state_dict().items()}
    ozq, oidx = VO.quant(wv, "mead", lat.cpu(), emo.cpu())
    assert torch.equal(info[2].cpu(), oidx)                       # slice-local indices of the per-clip emotion codebook
    verts = ae.decode(quanted)
    assert verts.shape == (2, L, 15069)
    assert mad(verts, VO.decode(wv, "mead", ozq)) < 1e-4


def test_biwi_demo_end_to_end(tmp_path):
    """demo_biwi.py wiring: wav -> wav2vec2-base -> pair frames -> BIWI denoiser (head_dim 256) -> DDIM -> quant
    (256 x 128 codebook) -> decode to 70110 coordinates."""
    from scipy.io import wavfile
    from fdm_amd import pipeline
    wav = (np.random.default_rng(1).standard_normal(16000) * 3000).astype(np.int16)
    wp = str(tmp_path / "biwi.wav")
    wavfile.write(wp, 16000, wav)
    dst = pipeline.demo_main("biwi", ["--audio_file", wp, "--audio_path", str(tmp_path / "result"), "--ddim_steps", "3"])
    arr = np.load(dst)
    assert arr.shape == (1, 49, 70110) and np.isfinite(arr).all()


def test_bf16_module_surface_tracks_fp32(monkeypatch):
    """FDM_AMD_DTYPE=bf16 through the drop-in classes (bf16 step program with folded norm3, bf16 HuBERT and VQ
    decoder) stays within the stated distance of the fp32 path on a short DDIM run."""
    from fdm_amd import pipeline
    gen = torch.Generator().manual_seed(9)
    wav = HO.processor_normalize(torch.randn(24000, generator=gen) * 0.1).numpy()
    res = {}
    for name in ("fp32", "bf16"):
        monkeypatch.setenv("FDM_AMD_DTYPE", name)
        diffusion, ae = pipeline.build_models("vocaset", device=DEV)
        out, latent = pipeline.animate(diffusion, ae, wav, ddim_steps=10, seed=4, device=DEV)
        assert torch.isfinite(out).all()
        res[name] = (out.cpu(), latent.cpu())
    assert res["fp32"][0].shape == (1, 74, 15069)
    assert mad(res["fp32"][1], res["bf16"][1]) < 0.15          # latents O(4)


def test_plans_survive_callers_inference_mode_blocks():
    """The reference decorates its samplers with @torch.inference_mode(); a plan first used inside such a block must keep
    working outside it (its workspaces and counters are refilled in place on every call)."""
    from fdm_amd.denoiser import DenoiserPlan
    from fdm_amd._lib import F32
    from oracle import weights as W
    inp = W.synth_inputs("vocaset_tiny", 2, 16, seed=5)
    with torch.inference_mode():
        plan = DenoiserPlan("vocaset_tiny", W.make_fdm_weights("vocaset_tiny"), F32, "cuda:0")
        plan.prepare(inp["hub"], inp["style"], L=16)
        a = plan.sample_ddpm(inp["x"].to("cuda:0"), [5, 4, 3, 2, 1, 0], seed=3).clone()
    b = plan.sample_ddpm(inp["x"].to("cuda:0"), [5, 4, 3, 2, 1, 0], seed=3)
    plan.prepare(inp["hub"][:1], inp["style"][:1], L=16)
    with torch.inference_mode():
        c = plan.sample_ddpm(inp["x"][:1].to("cuda:0"), [5, 4, 3, 2, 1, 0], seed=3)
    assert torch.equal(a, b) and torch.equal(c[0], a[0])


def test_checkpoint_files_round_trip(tmp_path):
    """stage-1 / stage-2 checkpoint FILES in the reference's layout ({'state_dict': ...} for the VQ-VAE, {'model': ...} for the
    diffusion model, samples/sample_diffusion_vocaset.py:26,91-97) written with torch.save and loaded through
    build_models(stage1=, stage2=): same state dicts, same animation bit for bit."""
    import numpy as np
    from fdm_amd import pipeline
    d0, ae0 = pipeline.build_models("vocaset", device=DEV)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for prm in list(d0.parameters()) + list(ae0.parameters()):
            if prm.dim() >= 2 and "audio_encoder" not in str(type(prm)):
                prm.add_(torch.randn(prm.shape, generator=g) * 1e-3)
    d0.denoise_fn._plan_stale = True
    ae0._plan_stale = True
    s1, s2 = str(tmp_path / "stage1.pth.tar"), str(tmp_path / "stage2.mpt")
    torch.save({"state_dict": ae0.state_dict()}, s1)
    torch.save({"model": d0.state_dict()}, s2)
    d1, ae1 = pipeline.build_models("vocaset", device=DEV, stage1=s1, stage2=s2)
    for a, b in ((d0.state_dict(), d1.state_dict()), (ae0.state_dict(), ae1.state_dict())):
        assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    wav = pipeline.processor_normalize(np.sin(np.arange(16000) * 0.05).astype(np.float32), pad_seconds=0.2)
    o0, _ = pipeline.animate(d0, ae0, wav, ddim_steps=5, seed=3, device=DEV)
    o1, _ = pipeline.animate(d1, ae1, wav, ddim_steps=5, seed=3, device=DEV)
    assert o0.shape[0] == 1 and torch.isfinite(o0).all() and torch.equal(o0, o1)


def test_single_clip_models_set_the_plan_and_stay_inside_the_contract():
    """pipeline.build_models(single_clip=True) -- what the sampler entry points use at batch size 1 -- puts the K-slice setting on the
    denoiser's plan (and on every plan the module rebuilds); the animated clip stays within fp32 rounding of the plain setting."""
    from fdm_amd import pipeline
    from fdm_amd.modules import SINGLE_CLIP_PLAN
    wav = pipeline.processor_normalize((torch.randn(32000, generator=torch.Generator().manual_seed(3)) * 0.1).numpy(), pad_seconds=0)
    outs = []
    for single in (False, True):
        diffusion, ae = pipeline.build_models("mead", device=DEV, dtype="f16x3", cfg_level=2.5, single_clip=single)
        model = diffusion.denoise_fn.model
        v, lat = pipeline.animate(diffusion, ae, torch.from_numpy(wav).unsqueeze(0), seed=5, device=DEV)
        plan = model.plan(torch.device(DEV))
        assert (plan.get("ksplit.out"), plan.get("ksplit.ffn2")) == ((SINGLE_CLIP_PLAN["ksplit.out"], SINGLE_CLIP_PLAN["ksplit.ffn2"]) if single else (1, 1))
        outs.append(lat.cpu())
    assert torch.isfinite(outs[1]).all() and float((outs[0] - outs[1]).abs().max()) < 1e-3      # 1000 DDPM steps apart by fp32 rounding only

