"""Generate the committed golden fixtures by running the *reference* on CPU (container only).

    python tests/golden/make_golden.py [--only NAME ...]

Imports /root/reference through tests/golden/refshim.py, loads the name-keyed synthetic weights of
oracle/weights.py into the reference modules (strict name/shape check), runs the reference hot path
on seeded synthetic inputs and stores inputs-that-cannot-be-regenerated + expected outputs as
tests/golden/*.npz.  It also prints the max-abs difference between the reference and the oracle
restatement for every case, which is how the oracle was pinned.
Weights and synthetic inputs are NOT stored: they are regenerated from their seeds.
"""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import refshim  # noqa: E402

_ARGV = list(sys.argv)          # refshim resets sys.argv for the reference's argparse-as-config
refshim.install(hubert_layers=24)

from oracle import fdm_oracle as FO  # noqa: E402
from oracle import hubert_oracle as HO  # noqa: E402
from oracle import vq_oracle as VO  # noqa: E402
from oracle import weights as W  # noqa: E402

import transformers  # noqa: E402

VERS = np.array([f"torch {torch.__version__}", f"transformers {transformers.__version__}"])
torch.set_grad_enabled(False)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, versions=VERS, **arrs)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.0f} KiB)")


def mad(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())


class _HubStub(torch.nn.Module):
    """Stands in for FDM.audio_encoder: returns given features (denoiser cases are HuBERT-free)."""

    def __init__(self):
        super().__init__()
        self.hub = None

    def forward(self, *a, **k):
        return type("O", (), {"last_hidden_state": self.hub})()


def check_load(module, wd, prefix=""):
    sd = module.state_dict()
    for k, v in wd.items():
        assert prefix + k in sd, f"missing in reference state_dict: {prefix + k}"
        assert tuple(sd[prefix + k].shape) == tuple(v.shape), (k, sd[prefix + k].shape, v.shape)
    module.load_state_dict({prefix + k: v for k, v in wd.items()}, strict=False)


_REF_CACHE = {}


def ref_fdm(preset):
    """Reference FDM with synthetic named weights and a stubbed audio encoder."""
    if preset in _REF_CACHE:
        return _REF_CACHE[preset]
    import models.hubert as rh
    from transformers import HubertConfig
    full = rh.HubertModel.from_pretrained
    rh.HubertModel.from_pretrained = classmethod(lambda cls, *a, **k: cls(HubertConfig(
        hidden_size=64, num_hidden_layers=1, num_attention_heads=4, intermediate_size=64,
        feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True, attn_implementation="eager")))
    p = W.PRESETS[preset]
    if preset.startswith("vocaset"):
        from models.fdm_vocaset import FDM
        m = FDM(feature_dim=p["d"], n_head=p["n_head"], num_layers=p["n_layers"])
    else:
        from models.fdm_vqvae_mead import FDM
        m = FDM(feature_dim=p["d"], n_head=p["n_head"], num_layers=p["n_layers"])
    rh.HubertModel.from_pretrained = full
    m.audio_encoder = _HubStub()
    m.eval()
    wd = W.make_fdm_weights(preset)
    check_load(m, wd)
    _REF_CACHE[preset] = (m, wd)
    return m, wd


def ref_denoise(preset, m, hub, t, x, style, emo=None):
    """One B=1 reference FDM.forward call on clip tensors."""
    m.audio_encoder.hub = hub.unsqueeze(0)
    tt = torch.full((1,), int(t), dtype=torch.long)
    audio = torch.zeros(1, 16)
    if emo is None:
        return m(audio, tt, x.unsqueeze(0), style.unsqueeze(0))[0]
    return m(audio, tt, x.unsqueeze(0), emo.unsqueeze(0), style.unsqueeze(0))[0]


# ---------------------------------------------------------------------------------------------
def g1_schedule():
    from video_diffusion_pytorch.diffusion_BIWI_encoder_decoder import GaussianDiffusion
    d = GaussianDiffusion(torch.nn.Identity(), timesteps=1000, loss_type="l2")
    names = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
             "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
             "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
             "posterior_mean_coef1", "posterior_mean_coef2"]
    ours = FO.schedule_buffers(1000)
    out = {}
    for n in names:
        out[n] = getattr(d, n).numpy()
        print(f"  {n}: bit-equal={bool((getattr(d, n) == ours[n]).all())}")
    for steps in (3, 50, 100, 250):
        times = np.linspace(-1, 999, steps + 1).astype(np.int32)
        times = list(reversed(times.tolist()))
        out[f"ddim_pairs_{steps}"] = np.array(list(zip(times[:-1], times[1:])), dtype=np.int32)
        assert FO.ddim_time_pairs(steps) == list(zip(times[:-1], times[1:]))
    save("schedule", **out)


def g2_masks():
    from models.fdm_vocaset import init_biased_mask, PeriodicPositionalEncoding, PositionalEncoding, enc_dec_mask
    out = {}
    rows = [0, 1, 29, 30, 35, 59, 60, 61, 299, 599]
    out["rows"] = np.array(rows)
    for (h, per) in ((8, 30), (4, 30), (4, 25), (2, 30)):
        m = init_biased_mask(n_head=h, max_seq_len=600, period=per)
        o = FO.biased_mask(h, 600, per)
        print(f"  biased_mask({h},600,{per}) bit-equal={bool((m == o).all())}")
        out[f"mask_{h}_{per}"] = m[:, rows, :].numpy()
    ppe = PeriodicPositionalEncoding(1024, period=30).pe[0]
    o = FO.positional_table(1024, "periodic", 30, 630)
    print(f"  PPE bit-equal={bool((ppe == o).all())}")
    out["ppe_1024_rows"] = ppe[rows[:8] + [629]].numpy()
    pe = PositionalEncoding(512).pe[0]
    o = FO.positional_table(512, "sinus", 30, 600)
    print(f"  PE bit-equal={bool((pe[:600] == o).all())}")
    out["pe_512_rows"] = pe[rows].numpy()
    mm = enc_dec_mask("cpu", "vocaset", 5, 7)
    out["enc_dec_mask_5_7"] = mm.numpy()
    save("masks", **out)


STEP_CASES = [(7, 0), (30, 1), (31, 500), (100, 999), (100, 0)]


def g3_fdm_step(preset, cases=STEP_CASES):
    m, wd = ref_fdm(preset)
    out = {}
    for (L, t) in cases:
        inp = W.synth_inputs(preset, 1, L, seed=100 + L)
        emo = inp["emo"][0] if "emo" in inp else None
        t0 = time.time()
        ref = ref_denoise(preset, m, inp["hub"][0], t, inp["x"][0], inp["style"][0], emo)
        tr = {}
        ours = FO.fdm_forward_clip(wd, preset, inp["hub"][0], t, inp["x"][0], inp["style"][0], emo, trace=tr)
        fold = FO.fdm_forward_clip(wd, preset, inp["hub"][0], t, inp["x"][0], inp["style"][0], emo, folded=True)
        print(f"  {preset} L={L} t={t}: |ref-oracle|={mad(ref, ours):.3e} |ref-folded|={mad(ref, fold):.3e} "
              f"|ref|max={float(ref.abs().max()):.3f} ({time.time() - t0:.1f}s)")
        out[f"x0_L{L}_t{t}"] = ref.numpy()
        if L == 30:
            out[f"h0_L{L}_t{t}"] = tr["h0"].numpy()
            out[f"layer0_L{L}_t{t}"] = tr["layer0"].numpy()
    out["cases"] = np.array(cases)
    save(f"fdm_step_{preset}", **out)


class _Randn:
    """Injects noise: replaces torch.randn / randn_like with a queue (RNG parity)."""

    def __init__(self, queue):
        self.q = list(queue)

    def __enter__(self):
        self.r, self.rl = torch.randn, torch.randn_like
        torch.randn = lambda *a, **k: self.q.pop(0)
        torch.randn_like = lambda *a, **k: self.q.pop(0)
        return self

    def __exit__(self, *a):
        torch.randn, torch.randn_like = self.r, self.rl


def g4_chains(preset):
    """DDPM p_sample chains t=9..0 and 999..990, DDIM 3 / 50 steps, outputs after every step."""
    m, wd = ref_fdm(preset)
    if preset.startswith("vocaset"):
        from video_diffusion_pytorch.diffusion_BIWI_encoder_decoder import GaussianDiffusion
    else:
        from video_diffusion_pytorch.diffusion_mead_encoder_decoder import GaussianDiffusion
    diff = GaussianDiffusion(m, timesteps=1000, loss_type="l2").eval()
    buf = FO.schedule_buffers()
    out = {}
    L = 12
    inp = W.synth_inputs(preset, 1, L, seed=7)
    emo = inp["emo"] if "emo" in inp else None
    conds = (inp["style"],) if emo is None else (emo, inp["style"])
    m.audio_encoder.hub = inp["hub"]
    audio = torch.zeros(1, 16)
    g = torch.Generator().manual_seed(99)
    for name, ts in (("lo", list(range(9, -1, -1))), ("hi", list(range(999, 989, -1)))):
        noise = torch.randn(len(ts), *inp["x"].shape, generator=g)
        x = inp["x"].clone()
        rec = []
        for i, t in enumerate(ts):
            with _Randn([noise[i]]):
                x = diff.p_sample(x, torch.full((1,), t, dtype=torch.long), audio, *conds)
            rec.append(x.clone())
        orec = []
        den = lambda xx, tt: FO.fdm_forward(wd, preset, inp["hub"], tt, xx, inp["style"], emo)
        FO.p_sample_loop(den, inp["x"].clone(), noise, ts, buf, orec)
        print(f"  {preset} ddpm {name}: |ref-oracle| per step max={max(mad(a, b) for a, b in zip(rec, orec)):.3e}")
        out[f"ddpm_{name}_noise"] = noise.numpy()
        out[f"ddpm_{name}_steps"] = torch.stack(rec).numpy()
        out[f"ddpm_{name}_t"] = np.array(ts)
    if preset.startswith("vocaset"):
        for steps in (3, 50):
            rec = []
            orig = diff.predict_noise_from_start
            with _Randn([inp["x"].clone()] + [torch.zeros_like(inp["x"])] * (steps + 1)):
                final = diff.ddim_sample(audio, tuple(inp["x"].shape), inp["style"], steps)
            den = lambda xx, tt: FO.fdm_forward(wd, preset, inp["hub"], tt, xx, inp["style"], emo)
            ours = FO.ddim_sample(den, inp["x"].clone(), steps, buf)
            print(f"  {preset} ddim {steps}: |ref-oracle|={mad(final, ours):.3e} |x|max={float(final.abs().max()):.2f}")
            out[f"ddim_{steps}_final"] = final.numpy()
    out["L"] = np.array(L)
    save(f"chains_{preset}", **out)


def g5_cfg1():
    """cfg-1 end-to-end as written: 1 clip x 100 frames, DDIM 50 steps, HuBERT-large inside the loop."""
    import models.hubert as rh
    from models.fdm_vocaset import FDM
    from video_diffusion_pytorch.diffusion_BIWI_encoder_decoder import GaussianDiffusion
    t0 = time.time()
    m = FDM(feature_dim=1024).eval()
    wd = W.make_fdm_weights("vocaset")
    wh = W.make_hubert_weights(24)
    check_load(m, wd)
    m.audio_encoder.load_state_dict(wh, strict=True)
    diff = GaussianDiffusion(m, timesteps=1000, loss_type="l2").eval()
    g = torch.Generator().manual_seed(1)
    wav = HO.processor_normalize(torch.randn(32080, generator=g) * 0.1).unsqueeze(0)
    xT = torch.randn(1, 1600, 64, generator=g)
    sid = torch.eye(8)[2:3]
    print(f"  build {time.time() - t0:.1f}s")
    t0 = time.time()
    with _Randn([xT.clone()] + [torch.zeros_like(xT)] * 51):
        final = diff.ddim_sample(wav, (1, 1600, 64), sid, 50)
    tref = time.time() - t0
    print(f"  reference cfg-1 as written: {tref:.2f} s -> {100 / tref:.2f} frames/s")
    t0 = time.time()
    hub = HO.hubert_forward(wh, wav, 24)
    den = lambda xx, tt: FO.fdm_forward(wd, "vocaset", hub, tt, xx, sid, None, folded=True)
    ours = FO.ddim_sample(den, xT.clone(), 50)
    print(f"  oracle hoisted+folded: {time.time() - t0:.2f} s; |ref-oracle|={mad(final, ours):.3e} "
          f"|x|max={float(final.abs().max()):.2f}")
    save("cfg1_e2e", final=final.numpy(), ref_seconds=np.array(tref), wav_seed=np.array(1))


def g6_hubert():
    import models.hubert as rh
    out = {}
    for layers in (2, 24):
        refshim.install(hubert_layers=layers)
        hm = rh.HubertModel.from_pretrained("x").eval()
        wh = W.make_hubert_weights(layers)
        hm.load_state_dict(wh, strict=True)
        for secs, n in ((2, 32000), (10, 160000)):
            if layers == 2 and secs == 10:
                continue
            g = torch.Generator().manual_seed(10 + secs)
            wav = HO.processor_normalize(torch.randn(n, generator=g) * 0.1)
            t0 = time.time()
            ref = hm(wav.unsqueeze(0), "vocaset").last_hidden_state[0]
            tr = {}
            ours = HO.hubert_forward_clip(wh, wav, layers, trace=tr)
            print(f"  hubert L{layers} {secs}s: out {tuple(ref.shape)} |ref-oracle|={mad(ref, ours):.3e} "
                  f"|ref|max={float(ref.abs().max()):.2f} ({time.time() - t0:.1f}s)")
            assert ref.shape[0] == HO.num_frames(n)
            if secs == 2:
                out[f"out_L{layers}_2s"] = ref.numpy()
                if layers == 2:
                    fe = hm.feature_extractor(wav.unsqueeze(0))[0].t()
                    fe = fe[: fe.shape[0] - fe.shape[0] % 2]
                    print(f"    conv stack |ref-oracle|={mad(fe, tr['conv']):.3e}")
                    out["conv_2s"] = fe.numpy()
            else:
                out[f"out_L{layers}_10s_rows8"] = ref[::8].numpy()
    refshim.install(hubert_layers=24)
    save("hubert", **out)


def g7_vq():
    from models.utils.config import vocaset_vq_vae_args, vq_vae_args, biwi_vq_vae_args
    from models.vq_vae_vocaset import VQAutoEncoder as V1
    from models.vq_vae_emotion import VQAutoEncoder as V2
    from models.vq_vae import VQAutoEncoder as V3
    out = {}
    for preset, V, args in (("vocaset", V1, vocaset_vq_vae_args()), ("mead", V2, vq_vae_args()),
                            ("biwi", V3, biwi_vq_vae_args())):
        p = W.PRESETS[preset]
        ae = V(args).eval()
        wd = W.make_vq_weights(preset)
        check_load(ae, wd)
        E = wd["quantize.embedding.weight"]
        for L in (2, 5, 12, 100):
            g = torch.Generator().manual_seed(40 + L)
            # latents near the codebook scale, plus exact-tie rows (z == a code) and mid-point rows
            z = torch.randn(1, L * p["G"], p["c"], generator=g) * (1.5 / 256)
            emos = range(p["n_books"]) if (p["n_books"] > 1 and L == 5) else [min(3, p["n_books"] - 1)]
            for e in emos:
                emo = torch.eye(7)[e] if p["n_books"] > 1 else None
                base = e * 256 if p["n_books"] > 1 else 0
                zz = z.clone()
                zz[0, 0] = E[base + 17]
                if zz.shape[1] > 2:
                    zz[0, 1] = 0.5 * (E[base + 3] + E[base + 200])
                    zz[0, 2] = E[base + 255]
                if p["n_books"] > 1:
                    zq, _, info = ae.quant(zz, emo)
                    ozq, oidx = VO.quant(wd, preset, zz, emo.unsqueeze(0))
                else:
                    zq, _, info = ae.quant(zz)
                    ozq, oidx = VO.quant(wd, preset, zz)
                idx = info[2]
                same = bool((idx == oidx).all())
                key = f"{preset}_L{L}_e{e}"
                out[key + "_idx"] = idx.numpy().astype(np.int16)
                if True:
                    t0 = time.time()
                    dec = ae.decode(zq)[0]
                    odec = VO.decode(wd, preset, ozq)[0]
                    print(f"  {key}: idx equal={same} |zq diff|={mad(zq, ozq):.1e} decode |ref-oracle|={mad(dec, odec):.3e} "
                          f"|dec|max={float(dec.abs().max()):.2f} ({time.time() - t0:.1f}s)")
                    out[key + "_zq_sum"] = np.array(float(zq.double().sum()))
                    if L <= 5 and e == emos[0] and preset != "biwi":
                        out[key + "_dec"] = dec.numpy()
                    else:
                        out[key + "_dec_cols16"] = dec[:, ::16].numpy()
    save("vq", **out)


def g8_cfg():
    preset = "mead"
    m, wd = ref_fdm(preset)
    out = {}
    L, t = 20, 321
    inp = W.synth_inputs(preset, 1, L, seed=55)
    a = ref_denoise(preset, m, inp["hub"][0], t, inp["x"][0], inp["style"][0], inp["emo"][0])
    u = ref_denoise(preset, m, inp["hub"][0], t, inp["x"][0], inp["style"][0], torch.zeros(7))
    # utiles/classifierfree.py:20-21
    scale = torch.ones(inp["x"].shape[1]) * 2.5
    mix = u + scale.view(-1, 1) * (a - u)
    ours = FO.fdm_forward_cfg(wd, preset, inp["hub"], t, inp["x"], inp["style"], inp["emo"], 2.5)[0]
    print(f"  cfg mix |ref-oracle|={mad(mix, ours):.3e}")
    save("cfg_mead", mix=mix.numpy(), cond=a.numpy(), uncond=u.numpy(), L=np.array(L), t=np.array(t))


def g9_audio():
    from transformers import Wav2Vec2FeatureExtractor
    from utiles.adaIN import adaptive_instance_normalization
    fe = Wav2Vec2FeatureExtractor(feature_size=1, sampling_rate=16000, padding_value=0.0,
                                  do_normalize=True, return_attention_mask=False)
    g = torch.Generator().manual_seed(3)
    wav = (torch.randn(8000, generator=g) * 0.1 + 0.01).numpy()
    ref = np.squeeze(fe(wav, sampling_rate=16000).input_values)
    ours = HO.processor_normalize(torch.from_numpy(wav))
    print(f"  processor normalise |ref-oracle|={mad(ref, ours):.3e}")
    c = torch.randn(2, 6, 11, generator=g)
    s = torch.randn(2, 6, 9, generator=g) * 2 + 1
    ar = adaptive_instance_normalization(c, s)
    print(f"  adaIN |ref-oracle|={mad(ar, FO.adain(c, s)):.3e}")
    save("audio_misc", wav=wav, normalized=ref.astype(np.float32), adain_c=c.numpy(), adain_s=s.numpy(),
         adain_out=ar.numpy())


def g10_state_keys():
    """Names + shapes of the reference state dicts (the checkpoint-compatibility surface)."""
    import json
    import models.hubert as rh
    from models.fdm_vocaset import FDM as F1
    from models.fdm_vqvae_mead import FDM as F2
    from video_diffusion_pytorch.diffusion_BIWI_encoder_decoder import GaussianDiffusion
    from models.utils.config import vocaset_vq_vae_args, vq_vae_args, biwi_vq_vae_args
    from models.vq_vae_vocaset import VQAutoEncoder as V1
    from models.vq_vae_emotion import VQAutoEncoder as V2
    from models.vq_vae import VQAutoEncoder as V3
    refshim.install(hubert_layers=24)
    out = {}
    sd = lambda m: {k: list(v.shape) for k, v in m.state_dict().items()}
    out["diffusion_vocaset"] = sd(GaussianDiffusion(F1(feature_dim=1024), timesteps=1000, loss_type="l2"))
    out["fdm_mead"] = sd(F2(feature_dim=512))
    out["vq_vocaset"] = sd(V1(vocaset_vq_vae_args()))
    out["vq_mead"] = sd(V2(vq_vae_args()))
    out["vq_biwi"] = sd(V3(biwi_vq_vae_args()))
    path = os.path.join(HERE, "state_keys.json")
    json.dump(out, open(path, "w"))
    print(f"  wrote state_keys.json ({os.path.getsize(path) / 1024:.0f} KiB)", {k: len(v) for k, v in out.items()})


def g11_wav2vec():
    """wav2vec2-base (BIWI audio encoder, models/wav2vec.py) with random-init base config."""
    import models.wav2vec as rw
    from transformers import Wav2Vec2Config
    from oracle import wav2vec_oracle as WO
    out = {}
    for layers in (2, 12):
        cfg = Wav2Vec2Config(num_hidden_layers=layers, attn_implementation="eager")
        m = rw.Wav2Vec2Model(cfg).eval()
        ww = W.make_wav2vec_weights(layers)
        sd = m.state_dict()
        missing = [k for k in sd if k not in ww]
        assert not missing, missing
        m.load_state_dict(ww, strict=True)
        for secs, n in ((2, 32000), (10, 160000)):
            if layers == 2 and secs == 10:
                continue
            g = torch.Generator().manual_seed(20 + secs)
            wav = HO.processor_normalize(torch.randn(n, generator=g) * 0.1)
            ref = m(wav.unsqueeze(0)).last_hidden_state[0]
            ours = WO.wav2vec_forward_clip(ww, wav, layers)
            print(f"  wav2vec2 L{layers} {secs}s: out {tuple(ref.shape)} |ref-oracle|={mad(ref, ours):.3e} |ref|max={float(ref.abs().max()):.2f}")
            if secs == 2:
                out[f"out_L{layers}_2s"] = ref.numpy()
            else:
                out[f"out_L{layers}_10s_rows8"] = ref[::8].numpy()
    save("wav2vec", **out)


def g15_vq_stats():
    """The whole return tuple of the reference's quant(): emb_loss, perplexity and the code histogram of min_encodings
    (its column sums; the one-hot matrix itself is the indices of vq.npz) -- VectorQuantizer.forward, quantizer.py:35-64."""
    from models.utils.config import vocaset_vq_vae_args, vq_vae_args, biwi_vq_vae_args
    from models.vq_vae_vocaset import VQAutoEncoder as V1
    from models.vq_vae_emotion import VQAutoEncoder as V2
    from models.vq_vae import VQAutoEncoder as V3
    out = {}
    for preset, V, args in (("vocaset", V1, vocaset_vq_vae_args()), ("mead", V2, vq_vae_args()), ("biwi", V3, biwi_vq_vae_args())):
        p = W.PRESETS[preset]
        ae = V(args).eval()
        wd = W.make_vq_weights(preset)
        check_load(ae, wd)
        for L in (5, 100):
            g = torch.Generator().manual_seed(140 + L)
            z = torch.randn(1, L * p["G"], p["c"], generator=g) * (1.5 / 256)
            for e in ([0, 5] if p["n_books"] > 1 else [0]):
                emo = torch.eye(7)[e] if p["n_books"] > 1 else None
                with torch.no_grad():
                    zq, loss, info = ae.quant(z, emo) if p["n_books"] > 1 else ae.quant(z)
                ol, op_, ome = VO.quant_stats(wd, preset, z, None if emo is None else emo.unsqueeze(0))
                key = f"{preset}_L{L}_e{e}"
                print(f"  {key}: loss {float(loss):.6e} (oracle {float(ol):.6e})  perplexity {float(info[0]):.5f} (oracle {float(op_):.5f})  "
                      f"min_encodings equal={bool(torch.equal(info[1], ome))}")
                out[key + "_loss"] = np.array(float(loss), dtype=np.float32)
                out[key + "_perplexity"] = np.array(float(info[0]), dtype=np.float32)
                out[key + "_hist"] = info[1].sum(0).numpy().astype(np.int32)
                out[key + "_idx"] = info[2].numpy().astype(np.int16)
                assert tuple(info[1].shape) == (z.shape[1], 256)
    save("vq_stats", **out)


def g12_vq_encode():
    """VQ-VAE encoders: encode -> quant -> decode round trip on the reference."""
    from models.utils.config import vocaset_vq_vae_args, vq_vae_args, biwi_vq_vae_args
    from models.vq_vae_vocaset import VQAutoEncoder as V1
    from models.vq_vae_emotion import VQAutoEncoder as V2
    from models.vq_vae import VQAutoEncoder as V3
    out = {}
    for preset, V, args in (("vocaset", V1, vocaset_vq_vae_args()), ("mead", V2, vq_vae_args()), ("biwi", V3, biwi_vq_vae_args())):
        p = W.PRESETS[preset]
        ae = V(args).eval()
        wd = W.make_vq_weights(preset, encoder=True)
        sd = ae.state_dict()
        assert set(k for k in sd if not k.endswith(".pe")) == set(wd), set(sd) ^ set(wd)
        ae.load_state_dict(wd, strict=False)
        L = 10
        g = torch.Generator().manual_seed(60)
        x = torch.randn(1, L, p["V3"], generator=g) * 0.3    # O(1) mapped features: keeps InstanceNorm well conditioned next to the emotion offset
        emo = torch.eye(7)[5] if p["n_books"] > 1 else None
        if emo is not None:
            h = ae.encode(x, emo)
            oh = VO.encode(wd, preset, x, emo.unsqueeze(0))
            zq, _, info = ae.quant(h, emo)
        else:
            h = ae.encode(x)
            oh = VO.encode(wd, preset, x)
            zq, _, info = ae.quant(h)
        dec = ae.decode(zq)[0]
        print(f"  {preset}: encode |ref-oracle|={mad(h, oh):.3e} |h|max={float(h.abs().max()):.2f}")
        out[f"{preset}_h"] = h[0].numpy()
        out[f"{preset}_idx"] = info[2].numpy().astype(np.int16)
        out[f"{preset}_dec_cols16"] = dec[:, ::16].numpy()
    save("vq_encode", **out)


def g13_metrics():
    """Evaluation metrics: run the reference's own computer_metrix.main() and
    compute_diversity() on a seeded synthetic dataset written to a temp dir; keep the numbers it prints."""
    import contextlib
    import io
    import json
    import pickle
    import re
    import tempfile
    import computer_metrix as CM
    from oracle import metrics_oracle as MO
    res = {}
    for dataset, nv, sentences, subjects in (("vocaset", 6172, [str(i) for i in range(46, 51)], ["FaceTalk_A", "FaceTalk_B"]),
                                             ("BIWI", 23370, ["e" + str(i).zfill(2) for i in range(37, 41)], ["F2", "M3"])):
        seed, frames = (7 if dataset == "vocaset" else 9), 11
        rs = np.random.RandomState(seed + 100)
        mouth = sorted(rs.choice(nv, 300, replace=False).tolist())
        upper = sorted(rs.choice(nv, 200, replace=False).tolist())
        with tempfile.TemporaryDirectory() as td:
            os.makedirs(os.path.join(td, "gt")); os.makedirs(os.path.join(td, "pred")); os.makedirs(os.path.join(td, "regions"))
            templates, seqs = {}, {}
            for si, subj in enumerate(subjects):
                tmpl, ss = MO.synth_sequences(seed + si, len(sentences), frames, nv)
                templates[subj] = tmpl.reshape(-1)
                for sent, (gt, pred) in zip(sentences, ss):
                    np.save(os.path.join(td, "gt", f"{subj}_{sent}.npy"), gt.reshape(gt.shape[0], -1))
                    np.save(os.path.join(td, "pred", f"{subj}_{sent}.npy"), pred.reshape(pred.shape[0], -1))
                    # a second prediction "conditioned on" the other subject, for compute_diversity
                    np.save(os.path.join(td, "pred", f"{subj}_{sent}_condition_{subjects[0]}.npy"), pred.reshape(pred.shape[0], -1))
                    np.save(os.path.join(td, "pred", f"{subj}_{sent}_condition_{subjects[1]}.npy"), gt.reshape(gt.shape[0], -1))
                    seqs[(subj, sent)] = (gt, pred)
            with open(os.path.join(td, "templates.pkl"), "wb") as f:
                pickle.dump(templates, f)
            if dataset == "BIWI":
                open(os.path.join(td, "regions", "lve.txt"), "w").write(", ".join(map(str, mouth)))
                open(os.path.join(td, "regions", "fdd.txt"), "w").write(", ".join(map(str, upper)))
            else:
                mm = np.zeros(nv); mm[mouth] = 0.5
                um = np.zeros(nv); um[upper] = 0.9
                open(os.path.join(td, "regions", "weighted_mouth_mask.txt"), "w").write("\n".join(f"{v:.3f}" for v in mm) + "\n")
                open(os.path.join(td, "regions", "forehead_mask.txt"), "w").write("\n".join(f"{v:.3f}" for v in um) + "\n")
            argv = ["x", "--train_subjects", " ".join(subjects), "--pred_path", os.path.join(td, "pred"),
                    "--gt_path", os.path.join(td, "gt"), "--region_path", os.path.join(td, "regions"),
                    "--templates_path", os.path.join(td, "templates.pkl"), "--dataset", dataset]
            buf = io.StringIO()
            old = sys.argv
            try:
                with contextlib.redirect_stdout(buf):
                    sys.argv = argv
                    CM.main()
                    sys.argv = argv + ["--test_subjects", " ".join(subjects)]
                    CM.compute_diversity()
            finally:
                sys.argv = old
        txt = buf.getvalue()
        grab = lambda label: float(re.findall(r"^" + re.escape(label) + r": ([-+0-9.e]+)$", txt, re.M)[-1])
        r = dict(seed=seed, frames=frames, nv=nv, subjects=subjects, sentences=sentences, mouth=mouth, upper=upper,
                 frame_number=int(re.findall(r"Frame Number: (\d+)", txt)[0]),
                 mean_vertex_error=grab("Mean Vertex Error"), lip_vertex_error=grab("Lip Vertex Error"),
                 fdd=grab("FDD"), abs_fdd=grab("ABS FDD"), diversity=grab("Diversity"))
        # oracle restatement on the same data
        gts = np.concatenate([seqs[(a, b)][0] for a in subjects for b in sentences])
        prs = np.concatenate([seqs[(a, b)][1] for a in subjects for b in sentences])
        fd = [MO.fdd(seqs[(a, b)][0], seqs[(a, b)][1], templates[a], upper) for a in subjects for b in sentences]
        mine = dict(mean_vertex_error=MO.mean_vertex_error(gts, prs), lip_vertex_error=MO.max_vertex_error(gts, prs, mouth),
                    fdd=sum(fd) / len(fd), abs_fdd=sum(abs(v) for v in fd) / len(fd))
        for k, v in mine.items():
            print(f"  {dataset} {k}: reference prints {r[k]:.4e}, oracle {v:.6e}")
            assert abs(v - r[k]) <= 6e-5 * abs(r[k]) + 1e-12, (k, v, r[k])
        res[dataset] = r
    json.dump(res, open(os.path.join(HERE, "metrics.json"), "w"))
    print("  wrote metrics.json")


def g14_hubert_frames():
    """frame_num crop before the encoder (models/hubert.py:97-98) on the reference HuBERT (2 layers), and the
    reference's linear_interpolation (models/hubert.py:62-69) on random features."""
    import models.hubert as rh
    out = {}
    refshim.install(hubert_layers=2)
    hm = rh.HubertModel.from_pretrained("x").eval()
    wh = W.make_hubert_weights(2)
    hm.load_state_dict(wh, strict=True)
    g = torch.Generator().manual_seed(12)
    wav = HO.processor_normalize(torch.randn(32000, generator=g) * 0.1)
    ref = hm(wav.unsqueeze(0), frame_num=20).last_hidden_state[0]
    ours = HO.hubert_forward_clip(wh, wav, 2, frame_num=20)
    print(f"  frame_num=20: out {tuple(ref.shape)} |ref-oracle|={mad(ref, ours):.3e}")
    out["out_L2_2s_fn20"] = ref.numpy()
    refshim.install(hubert_layers=24)
    for (T, To, C) in ((99, 60, 512), (498, 299, 64), (7, 20, 16), (5, 1, 8)):
        x = torch.randn(2, T, C, generator=g)
        y = rh.linear_interpolation(x, 50, 30, output_len=To)
        mine = HO.linear_interpolation(x, 50, 30, output_len=To)
        print(f"  linear_interpolation {T}->{To}: |ref-oracle|={mad(y, mine):.3e}")
        out[f"interp_{T}_{To}_{C}_x"] = x.numpy()
        out[f"interp_{T}_{To}_{C}_y"] = y.numpy()
    y = rh.linear_interpolation(x, 50, 30)
    assert y.shape[1] == HO.linear_interpolation(x, 50, 30).shape[1]
    save("hubert_frames", **out)


ALL = {
    "schedule": g1_schedule, "masks": g2_masks,
    "fdm_step_vocaset": lambda: g3_fdm_step("vocaset"),
    "fdm_step_mead": lambda: g3_fdm_step("mead"),
    "fdm_step_vocaset_tiny": lambda: g3_fdm_step("vocaset_tiny", [(7, 0), (30, 1), (33, 500), (64, 999)]),
    "fdm_step_mead_tiny": lambda: g3_fdm_step("mead_tiny", [(7, 0), (30, 1), (33, 500), (64, 999)]),
    "chains_vocaset": lambda: g4_chains("vocaset"),
    "chains_mead": lambda: g4_chains("mead"),
    "chains_vocaset_tiny": lambda: g4_chains("vocaset_tiny"),
    "cfg1_e2e": g5_cfg1, "hubert": g6_hubert, "vq": g7_vq, "cfg_mead": g8_cfg, "audio_misc": g9_audio, "state_keys": g10_state_keys, "wav2vec": g11_wav2vec, "vq_encode": g12_vq_encode, "vq_stats": g15_vq_stats,
    "metrics": g13_metrics, "hubert_frames": g14_hubert_frames,
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args(_ARGV[1:])
    torch.set_num_threads(8)
    for name, fn in ALL.items():
        if a.only and name not in a.only:
            continue
        print(f"[{name}]")
        fn()
