"""End-to-end clip pipeline: wav -> processor normalisation -> HuBERT -> T-step sampling -> quant ->
decode (+ template) -> [B, L, V3] vertices, as wired by the reference's coherent callers
(samples/sample_diffusion_vocaset.py:59-88, samples/sample_diffusion_mead.py:67-86) and intended by
demo/demo_*.py:77-106 (flags + I/O layout kept; the demos' undefined names are re-wired per samples/)."""
import os

import numpy as np
import torch

from . import presets
from .modules import (FDM, ClassifierFreeSampleModel, FDMBiwi, FDMMead, GaussianDiffusion, VQAutoEncoder)

EMOTIONS = ["angry", "contempt", "disgusted", "fear", "happy", "sad", "surprised"]     # demo/demo_3d_mead.py:118


def load_wav(path, sr=16000):
    """16 kHz mono float32 (librosa.load(sr=16000) of the demos; scipy-based, no librosa here)."""
    from scipy.io import wavfile
    from scipy.signal import resample_poly
    rate, x = wavfile.read(path)
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(np.iinfo(x.dtype).max + 1)
    elif x.dtype.kind == "u":
        x = (x.astype(np.float32) - 128.0) / 128.0
    x = x.astype(np.float32)
    if x.ndim > 1:
        x = x.mean(axis=1)
    if rate != sr:
        g = np.gcd(int(rate), sr)
        x = resample_poly(x, sr // g, int(rate) // g).astype(np.float32)
    return x


def processor_normalize(x, pad_seconds=1.0, sr=16000):
    """Wav2Vec2Processor default: zero mean / unit variance ((x - mu)/sqrt(var + 1e-7)), then the demo's
    1 s of trailing zeros (demo/demo_vocaset.py:84-90)."""
    x = np.asarray(x, dtype=np.float32)
    x = (x - x.mean()) / np.sqrt(x.var() + 1e-7)
    if pad_seconds:
        x = np.concatenate([x, np.zeros(int(pad_seconds * sr), dtype=np.float32)])
    return x.astype(np.float32)


def build_models(preset="vocaset", feature_dim=None, device="cuda:0", stage1=None, stage2=None, dtype=None, cfg_level=None, single_clip=False):
    """(diffusion, autoencoder) with reference-compatible state dicts; checkpoints are loaded when the
    files exist ('model' / 'state_dict' keys as samples/sample_diffusion_vocaset.py:26,91-97), otherwise the
    seeded random init is kept (there are no checkpoints in this environment).  single_clip: the caller samples one clip per call
    (the reference's batch size) -- the step program's single-clip setting (modules.SINGLE_CLIP_PLAN)."""
    from dropin_config import vq_args_for
    p = presets.get(preset)
    cls = {"vocaset": FDM, "mead": FDMMead, "biwi": FDMBiwi}[p.name]
    kw = dict(feature_dim=feature_dim or p.d, n_head=(feature_dim or p.d) // p.head_dim, dtype=dtype)
    model = cls(**kw)
    if single_clip:
        from .modules import SINGLE_CLIP_PLAN
        for k, v in SINGLE_CLIP_PLAN.items():
            model.set_plan_option(k, v)
    ae = VQAutoEncoder(vq_args_for(p.name), dtype=dtype)
    denoise = ClassifierFreeSampleModel(model, cfg_level) if cfg_level else model
    diffusion = GaussianDiffusion(denoise, timesteps=1000, loss_type="l2")
    if stage2 and os.path.exists(stage2):
        diffusion.load_state_dict(torch.load(stage2, map_location="cpu")["model"], strict=False)
    else:
        # untrained latent_decoder is zero-initialised in the reference (outputs would be identically 0):
        # give the synthetic model a non-trivial head so the pipeline is exercised end to end
        g = torch.Generator().manual_seed(7)
        model.latent_decoder.weight.data.copy_(torch.randn(model.latent_decoder.weight.shape, generator=g) * 0.02)
    if stage1 and os.path.exists(stage1):
        ck = torch.load(stage1, map_location="cpu")
        ae.load_state_dict(ck.get("state_dict", ck.get("model", ck)), strict=False)
    return diffusion, ae


def _clip_x_T(shape, S, seed):
    """DDPM start of a sampling call: one x_T per CLIP from a CPU generator seeded with `seed` (the DDIM branch draws it the
    same way), shared by the clip's S conditions -- so `seed=` reproduces a call, and the sequential S = 1 calls of a style loop
    start where the batched call starts."""
    x = torch.randn(shape, generator=torch.Generator(device="cpu").manual_seed(seed))
    return x.repeat_interleave(S, dim=0) if S > 1 else x


@torch.no_grad()
def animate(diffusion, autoencoder, audio, template=None, id_one_hot=None, emotion_one_hot=None, steps=None,
            ddim_steps=None, seed=0, device="cuda:0"):
    """audio [B, n] (processor-normalised) -> vertices [B, L, V3].  DDPM full chain by default, DDIM if ddim_steps.

    id_one_hot [B*S, n_style] (and emotion_one_hot [B*S, n_emo]) with S > 1 animates every clip under S conditions in ONE
    sampling call -- the reference sampler's style loop (samples/sample_diffusion_vocaset.py:71-83) batched: the audio
    encoder and the audio tables run once per clip, the S conditions ride the same step program.  Returns [B*S, L, V3] in
    (clip, condition) order.  x_T is drawn per CLIP from a CPU generator seeded with `seed` on every path (S = 1 too), so a call is
    reproducible from its seed and every condition of a clip starts from the clip's x_T -- what S sequential calls with the same
    seed do.  DDIM (eta = 0: x_T is the only random input): bit-identical to the sequential loop.  DDPM: the per-step noise is
    Philox keyed by (seed, ROW = clip * S + condition, step), so a sequential call (row 0) draws the stream of the batched call's
    row 0 only: the other conditions differ from their sequential calls by that stream -- equal in distribution (the reference's
    sequential calls draw from one running generator), not bit for bit."""
    model = diffusion.denoise_fn.model if isinstance(diffusion.denoise_fn, ClassifierFreeSampleModel) else diffusion.denoise_fn
    p = model.preset
    audio = torch.as_tensor(audio, dtype=torch.float32, device=device)
    if audio.dim() == 1:
        audio = audio.unsqueeze(0)
    B = audio.shape[0]
    if id_one_hot is None:
        id_one_hot = torch.eye(p.n_style)[:1].expand(B, -1)
    id_one_hot = id_one_hot.reshape(-1, id_one_hot.shape[-1]).to(device)
    rows = max(id_one_hot.shape[0], 1 if emotion_one_hot is None else emotion_one_hot.reshape(-1, emotion_one_hot.shape[-1]).shape[0], B)
    if rows % B:
        raise ValueError(f"{rows} condition rows for {B} clips")
    S = rows // B
    hub = model.audio_features(audio)
    L = min(hub.shape[1] // p.pair, p.max_len)        # samples/sample_diffusion_vocaset.py:76 (no interpolation, a17b)
    shape = (B, L * p.G, p.c)
    if p.n_emo:
        if emotion_one_hot is None:
            emotion_one_hot = torch.eye(p.n_emo)[4:5].expand(rows, -1)
        emotion_one_hot = emotion_one_hot.reshape(-1, emotion_one_hot.shape[-1]).to(device)
        if emotion_one_hot.shape[0] == 1 and rows > 1:
            emotion_one_hot = emotion_one_hot.expand(rows, -1)
        if id_one_hot.shape[0] == 1 and rows > 1:
            id_one_hot = id_one_hot.expand(rows, -1)
        latent = diffusion.sample(audio, shape, emotion_one_hot, id_one_hot, seed=seed, x_T=_clip_x_T(shape, S, seed))
        quanted, _, _ = autoencoder.quant(latent, emotion_one_hot, stats=False)
    else:
        if ddim_steps:
            g = torch.Generator(device="cpu").manual_seed(seed)
            x_T = torch.randn(shape, generator=g).repeat_interleave(S, dim=0)
            latent = diffusion.ddim_sample(audio, shape, id_one_hot, ddim_steps, x_T=x_T)
        else:
            latent = diffusion.sample(audio, shape, id_one_hot, seed=seed, x_T=_clip_x_T(shape, S, seed))
        quanted, _, _ = autoencoder.quant(latent, stats=False)
    out = autoencoder.decode(quanted)
    if template is not None:
        tp = torch.as_tensor(template, dtype=torch.float32, device=device).reshape(-1, 1, out.shape[-1])
        out = out + (tp.repeat_interleave(S, dim=0) if (S > 1 and tp.shape[0] == B and B > 1) else tp)
    return out, latent


@torch.no_grad()
def animate_many(diffusion, autoencoder, audios, templates=None, id_one_hots=None, emotion_one_hots=None, ddim_steps=None,
                 seed=0, device="cuda:0", max_batch=8, bucket=16):
    """A test set's clips (different durations) through ONE sampling call per group of `max_batch` clips.

    The reference's samplers take the clips of a loader one at a time (bs = 1: samples/sample_diffusion_vocaset.py:51,71-83),
    which is the few-hundred-row regime where this path reaches 1-2 % of the MFMA roofline.  Clips of different lengths batch
    EXACTLY: the denoiser's self-attention is causal (models/fdm_vocaset.py:85,107-115: key j > query i is masked), every other
    op is per row, and the in-kernel noise is keyed by (clip, element index inside the clip) -- so a clip padded at its END to
    the group's longest length computes, for its own frames, bit for bit what its B = 1 call computes; the tail rows are
    discarded.  The audio encoder and the VQ decoder are not causal: they run per clip at the clip's own length (once each, < 1 %
    of the job).  audios: list of processor-normalised waveforms [n_b]; returns (list of [1, L_b, V3] vertices, list of latents)
    in the caller's order.  DDIM (noise-free): clips are grouped by length (least padding).  DDPM: groups follow the caller's
    order and clip b draws the noise stream of index b (Philox key clip0 + position), so results do not depend on max_batch.
    A group's length is rounded up to a multiple of `bucket` frames (free: the padding is exact), so a long-running caller
    cycles through a handful of shapes whose recorded step programs and tuned tiles the plan keeps."""
    model = diffusion.denoise_fn.model if isinstance(diffusion.denoise_fn, ClassifierFreeSampleModel) else diffusion.denoise_fn
    p = model.preset
    n = len(audios)
    wavs = [torch.as_tensor(a, dtype=torch.float32, device=device).reshape(1, -1) for a in audios]
    hubs = [model.audio_encoder(w).last_hidden_state for w in wavs]                 # [1, N_b, fw] each, own length
    Ls = [min(h.shape[1] // p.pair, p.max_len) for h in hubs]

    def row(x, b, width, default):
        if x is None:
            return default
        x = torch.as_tensor(x[b] if isinstance(x, (list, tuple)) else x, dtype=torch.float32).reshape(-1, width)
        return x[b:b + 1] if x.shape[0] == n else x[:1]
    ddim = bool(ddim_steps) and not p.n_emo
    order = sorted(range(n), key=lambda b: Ls[b]) if ddim else list(range(n))
    verts, lats = [None] * n, [None] * n
    prev = (model._hub_key, model._hub)
    try:
        for g0 in range(0, n, max_batch):
            grp = order[g0:g0 + max_batch]
            Lmax = max(Ls[b] for b in grp)
            if bucket and bucket > 1:
                Lmax = min(-(-Lmax // bucket) * bucket, p.max_len)
            Nmax = max(max(hubs[b].shape[1] for b in grp), Lmax * p.pair)
            hub = torch.zeros(len(grp), Nmax, hubs[grp[0]].shape[2], device=device)
            x_T = torch.zeros(len(grp), Lmax * p.G, p.c)
            for i, b in enumerate(grp):
                hub[i, :hubs[b].shape[1]] = hubs[b][0]
                gen = torch.Generator(device="cpu").manual_seed(seed)                 # what animate() draws for this clip alone
                x_T[i, :Ls[b] * p.G] = torch.randn((1, Ls[b] * p.G, p.c), generator=gen)[0]
            ids = torch.cat([row(id_one_hots, b, p.n_style, torch.eye(p.n_style)[:1]) for b in grp]).to(device)
            model.set_audio_features(hub)
            dummy = torch.zeros(len(grp), 1, device=device)
            shape = (len(grp), Lmax * p.G, p.c)
            if p.n_emo:
                emos = torch.cat([row(emotion_one_hots, b, p.n_emo, torch.eye(p.n_emo)[4:5]) for b in grp]).to(device)
                lat = diffusion.sample(dummy, shape, emos, ids, seed=seed, x_T=x_T, clip0=g0)
            elif ddim_steps:
                lat = diffusion.ddim_sample(dummy, shape, ids, ddim_steps, x_T=x_T)
            else:
                lat = diffusion.sample(dummy, shape, ids, seed=seed, x_T=x_T, clip0=g0)
            for i, b in enumerate(grp):
                lb = lat[i:i + 1, :Ls[b] * p.G].contiguous()
                q = autoencoder.quant(lb, emos[i:i + 1], stats=False)[0] if p.n_emo else autoencoder.quant(lb, stats=False)[0]
                out = autoencoder.decode(q)
                if templates is not None:
                    tp = templates[b] if isinstance(templates, (list, tuple)) else templates
                    out = out + torch.as_tensor(tp, dtype=torch.float32, device=device).reshape(1, 1, -1)
                verts[b], lats[b] = out, lb
    finally:
        model._hub_key, model._hub = prev
    return verts, lats


def demo_main(preset, argv=None):
    """CLI of demo/demo_{vocaset,biwi,3d_mead}.py:109-121: same flags, output = np.save(<audio_path>/<stem>.npy, [1, L, V3])."""
    import argparse
    p = presets.get(preset)
    ap = argparse.ArgumentParser(description="Expressive 3D Facial Animation Generation Based on Local-to-global Latent Diffusion")
    ap.add_argument("--audio_file", type=str, help="the audio file path for prediction")
    if p.n_emo:
        ap.add_argument("--emotion", type=str, default="happy", choices=EMOTIONS)
    ap.add_argument("--vertice_dim", type=int, default=p.V3)
    ap.add_argument("--feature_dim", type=int, default=p.d)
    ap.add_argument("--device", type=str, default="cuda:0")
    ap.add_argument("--template_file", type=str, default="templates.pkl")
    ap.add_argument("--stage1_model_path", type=str, default=f"{p.name}/{p.name}_stage1.mpt")
    ap.add_argument("--stage2_model_path", type=str, default=f"{p.name}/{p.name}_stage2.mpt")
    ap.add_argument("--audio_path", type=str, default=f"{p.name}/result")
    ap.add_argument("--ddim_steps", type=int, default=0, help="build-added: DDIM steps (0 = full DDPM chain)")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)
    diffusion, ae = build_models(p.name, a.feature_dim, a.device, a.stage1_model_path, a.stage2_model_path,
                                 cfg_level=None)
    wav = processor_normalize(load_wav(a.audio_file))
    template = None
    if os.path.exists(a.template_file) and a.template_file.endswith(".npy"):
        template = np.load(a.template_file).reshape(1, -1)
    emo = None
    if p.n_emo:
        emo = torch.eye(p.n_emo)[EMOTIONS.index(a.emotion)].unsqueeze(0)
    out, _ = animate(diffusion, ae, wav, template, None, emo, ddim_steps=a.ddim_steps, seed=a.seed, device=a.device)
    os.makedirs(a.audio_path, exist_ok=True)
    dst = os.path.join(a.audio_path, os.path.basename(a.audio_file)[:-4])
    np.save(dst, out.detach().cpu().numpy())
    print(f"saved {dst}.npy {tuple(out.shape)}")
    return dst + ".npy"
