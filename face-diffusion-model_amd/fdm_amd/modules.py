"""Host-side mirror of the reference's Python class surface (the drop-in boundary).

Same class names, constructor arguments, method names, argument order, tensor shapes and state-dict
keys as the reference, so that a reference checkpoint loads and the reference's callers
(samples/sample_diffusion_*.py, demo/demo_*.py) run unchanged -- but every forward pass executes on
the HIP path (fdm_amd.denoiser / hubert / vq through libfdm_hip.so).  There is no CPU fallback:
calling forward on CPU tensors raises.

  FDM (VOCASET)          models/fdm_vocaset.py:8-91
  FDM (3D-MEAD)          models/fdm_vqvae_mead.py:8-104
  FDM (BIWI)             models/fdm.py:9-99        (struct='Dec' + regroup x8: build-defined)
  GaussianDiffusion      video_diffusion_pytorch/diffusion_BIWI_encoder_decoder.py:549-761,
                         video_diffusion_pytorch/diffusion_mead_encoder_decoder.py:641-671
  HubertModel            models/hubert.py:72-146
  VQAutoEncoder          models/vq_vae_vocaset.py:9-43, models/vq_vae_emotion.py:9-41, models/vq_vae.py
  ClassifierFreeSampleModel  utiles/classifierfree.py:8-21
"""
import math
import os
from dataclasses import replace
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops, presets, schedule, synth
from ._lib import BF16, DTYPE_NAMES, F16, F16X3, F32, FdmError
from .denoiser import DenoiserPlan
from .hubert import HUBERT_LARGE, WAV2VEC2_BASE, HubertPlan, num_frames
from .vq import VQPlan


def compute_dtype(name=None):
    """Arithmetic mode of the HIP path: "fp32" (default; exact fp32 MFMA), "f16x3" (split-fp16 operands on the 16-bit matrix
    cores: same 1e-4 contract, ~1.6x the frames/s), "bf16" (throughput mode, BASELINE.json configs[1]), "f16" (single-plane fp16: bf16's
    speed, 11 significand bits instead of 8 in the step program; encoder on split-fp16 operands, VQ stages in fp32)."""
    name = (name or os.environ.get("FDM_AMD_DTYPE", "fp32")).lower()
    name = {"fp32": "f32", "float32": "f32", "bfloat16": "bf16"}.get(name, name)
    if name not in DTYPE_NAMES:
        raise FdmError(f"unknown compute dtype {name!r} (one of {sorted(DTYPE_NAMES)})")
    return DTYPE_NAMES[name]


def _side_dtype(dt):
    """VQ quant / decode (once per clip) run in fp32 when the step program uses split operands.  (VQPlan has an f16x3 mode too --
    its transformers on split-fp16 operands, 2.5 ms instead of 3.4 ms per 4 x 498 frames -- but its decode sits at 1.4-4.6e-5 from
    the reference on O(12) vertices against 1.5e-5 in fp32: the drop-in pipeline keeps the wider margin.)"""
    return dt if dt in (F32, BF16) else F32


def _audio_dtype(dt):
    """The audio encoders follow the step program's mode where they have it: fp32, bf16, or f16x3 (split-fp16 transformer layers
    and conv front: inside the 1e-4 contract at about half the fp32 encoder's time)."""
    if dt == F16:      # single-plane fp16 is a mode of the step program only: the once-per-clip encoder runs on split-fp16 operands
        return F16X3
    return dt if dt in (F32, BF16, F16X3) else F32


class ParamTree(nn.Module):
    """Registers parameters/buffers under dotted names so state_dict keys equal the reference's."""

    def _register(self, name, tensor, buffer=False):
        mod = self
        parts = name.split(".")
        for part in parts[:-1]:
            if part not in mod._modules:
                mod.add_module(part, nn.Module())
            mod = mod._modules[part]
        if buffer:
            mod.register_buffer(parts[-1], tensor)
        else:
            mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))

    def _load_from_state_dict(self, *a, **k):
        self._plan_stale = True
        return super()._load_from_state_dict(*a, **k)


class _TensorKey:
    """Identity of a cached input: the tensor OBJECT (kept alive here, so the caching allocator cannot hand its address to
    another clip) plus its version counter (in-place writes invalidate the cache)."""

    def __init__(self, t):
        self.t, self.version = t, (None if t is None else t._version)

    def matches(self, t):
        return t is self.t and (t is None or t._version == self.version)


def _scalar_t(t):
    """The timestep a step program runs at: all of its rows share it.  Callers route mixed per-clip timesteps through
    _mixed_t / _rows_one_by_one first, so a mixed tensor here is a bug, not an input."""
    if not torch.is_tensor(t):
        return int(t)
    f = t.flatten()
    if f.numel() > 1 and bool((f != f[0]).any()):
        raise FdmError("internal: mixed timesteps reached a step program (they run one row block at a time)")
    return int(f[0])


def _mixed_t(t):
    """True for a [B] timestep tensor whose entries differ -- what the reference's signatures allow (t is a [B] tensor everywhere:
    diffusion_BIWI_encoder_decoder.py:665,690,759) although its models only run B = 1."""
    return torch.is_tensor(t) and t.numel() > 1 and bool((t.flatten() != t.flatten()[0]).any())


def _row(x, i, rows):
    """Row i of a per-row argument ([rows, ...] tensors; anything else is shared by all rows)."""
    return x[i:i + 1] if (torch.is_tensor(x) and x.dim() > 1 and x.shape[0] == rows) else x


def _rows_one_by_one(model, fn, t, rows, audio, *per_row):
    """fn(audio_i, t_i, *args_i) for every row block i with its own timestep, results concatenated: per-clip timesteps mean what
    they mean in the reference -- independent B = 1 calls (section 2 item 5 of DESIGN.md).  rows = B * S blocks over B clips."""
    tf = t.flatten()
    if tf.numel() != rows:
        raise FdmError(f"{tf.numel()} timesteps for {rows} row blocks")
    B = audio.shape[0] if torch.is_tensor(audio) else rows
    S = max(rows // max(B, 1), 1)
    inj = model._hub if getattr(model, "_hub_key", None) == "injected" else None
    outs = []
    try:
        for i in range(rows):
            b = i // S
            if inj is not None:
                model.set_audio_features(inj[b:b + 1])
            a_i = audio[b:b + 1] if (torch.is_tensor(audio) and audio.dim() > 1) else audio
            outs.append(fn(a_i, tf[i:i + 1], *[_row(x, i, rows) for x in per_row]))
    finally:
        if inj is not None:
            model.set_audio_features(inj)
    return torch.cat(outs)


# --------------------------------------------------------------------------------------------------
def linear_interpolation(features, input_fps, output_fps, output_len=None):
    """models/hubert.py:62-69: resample [B, T, C] features along time with F.interpolate(mode='linear',
    align_corners=True); output_len defaults to int(T / input_fps * output_fps).  HIP kernel (fdm_op_linear_interp)."""
    if not features.is_cuda:
        raise FdmError("linear_interpolation runs on the HIP path only: move the features to the GPU")
    B, T, Cn = features.shape
    if output_len is None:
        output_len = int(T / float(input_fps) * output_fps)
    x = features.detach().to(torch.float32).contiguous()
    y = torch.empty(B, output_len, Cn, device=x.device)
    ops.linear_interp(x, y, B, T, output_len, Cn)
    return y


class HubertModel(ParamTree):
    """HuBERT-large (24 layers) with the reference's forward override; `from_pretrained` loads a local
    HF checkpoint directory when present, else keeps the seeded random init (no network here)."""

    def __init__(self, config=None, n_layers=24, seed=0, dtype=None):
        super().__init__()
        self.n_layers = int(getattr(config, "num_hidden_layers", n_layers)) if config is not None else n_layers
        self.config = config or SimpleNamespace(num_hidden_layers=self.n_layers, hidden_size=1024, output_attentions=False)
        for k, v in synth.make_hubert_weights(self.n_layers, seed).items():
            self._register(k, v)
        self.feature_extractor._freeze_parameters = lambda: None      # models/fdm_vocaset.py:19
        self._dtype = compute_dtype(dtype)
        self._plan = None
        self._plan_stale = True

    @classmethod
    def from_pretrained(cls, path=None, *a, **k):
        """Loads a local HF checkpoint directory (no network here); a missing directory keeps the seeded random init and
        says so.  Accepts the `hubert.` / `wav2vec2.` key prefixes of the *ForCTC checkpoints and the torch-2.0
        weight-norm names (weight_g / weight_v) of the positional conv."""
        import warnings
        m = cls()
        loaded = 0
        if path and os.path.isdir(str(path)):
            for fn in ("model.safetensors", "pytorch_model.bin"):
                fp = os.path.join(str(path), fn)
                if os.path.exists(fp):
                    if fn.endswith(".safetensors"):
                        from safetensors.torch import load_file
                        sd = load_file(fp)
                    else:
                        sd = torch.load(fp, map_location="cpu")
                    loaded = m.load_hf_state_dict(sd)
                    break
        if not loaded:
            warnings.warn(f"{cls.__name__}.from_pretrained({path!r}): no checkpoint found, keeping the seeded random init")
        return m

    def load_hf_state_dict(self, sd):
        """Key-normalise and load; returns the number of tensors taken.  A shape mismatch raises (strict=False does not
        cover shapes), so a HuBERT checkpoint cannot be loaded into the wav2vec2-base encoder by accident."""
        out = {}
        for kk, vv in sd.items():
            for pre in ("hubert.", "wav2vec2."):
                if kk.startswith(pre):
                    kk = kk[len(pre):]
            kk = kk.replace("conv.weight_g", "conv.parametrizations.weight.original0").replace("conv.weight_v", "conv.parametrizations.weight.original1")
            out[kk] = vv
        own = self.state_dict()
        take = {kk: vv for kk, vv in out.items() if kk in own}
        for kk, vv in take.items():
            if tuple(vv.shape) != tuple(own[kk].shape):
                raise FdmError(f"checkpoint tensor {kk} has shape {tuple(vv.shape)}, this encoder expects {tuple(own[kk].shape)}")
        self.load_state_dict(take, strict=False)
        return len(take)

    def _get_plan(self, device):
        if self._plan is None or self._plan_stale or self._plan.device != torch.device(device):
            self._plan = HubertPlan(self.state_dict(), self.n_layers, _audio_dtype(self._dtype), device)
            self._plan_stale = False
        return self._plan

    def forward(self, input_values, attention_mask=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, frame_num=None, interp_fps=None):
        """models/hubert.py:75-146.  frame_num crops the conv features to 2*frame_num frames BEFORE the encoder
        (:97-98).  Build-added `interp_fps=(input_fps, output_fps)`: resample the conv features with
        linear_interpolation (to `frame_num` frames when given) instead of the even crop -- the CodeTalker-style
        'vocaset' branch the shipped file dropped; default None = shipped behaviour."""
        if not input_values.is_cuda:
            raise FdmError("HubertModel.forward runs on the HIP path only: move the audio to the GPU")
        # a `str` second positional argument (models/fdm_vocaset.py:59 passes 'vocaset') is ignored
        out = self._get_plan(input_values.device).forward(input_values, frame_num=frame_num, interp_fps=interp_fps)
        return SimpleNamespace(last_hidden_state=out, hidden_states=None, attentions=None)


class Wav2Vec2Model(HubertModel):
    """wav2vec2-base (12 layers, d = 768) with the reference's forward override (models/wav2vec.py:69-143):
    the BIWI denoiser's audio encoder.  Same HF key names; GroupNorm conv stack, post-LN encoder."""

    def __init__(self, config=None, n_layers=12, seed=0, dtype=None):
        ParamTree.__init__(self)
        self.n_layers = int(getattr(config, "num_hidden_layers", n_layers)) if config is not None else n_layers
        self.config = config or SimpleNamespace(num_hidden_layers=self.n_layers, hidden_size=768, output_attentions=False)
        for k, v in synth.make_wav2vec_weights(self.n_layers, seed).items():
            self._register(k, v)
        self.feature_extractor._freeze_parameters = lambda: None
        self._dtype = compute_dtype(dtype)
        self._plan = None
        self._plan_stale = True

    def _get_plan(self, device):
        if self._plan is None or self._plan_stale or self._plan.device != torch.device(device):
            self._plan = HubertPlan(self.state_dict(), self.n_layers, _audio_dtype(self._dtype), device, cfg=WAV2VEC2_BASE)
            self._plan_stale = False
        return self._plan


# --------------------------------------------------------------------------------------------------
# The single-clip setting of the step program: the reference's samplers and demos issue ONE clip per call (bs = 1,
# samples/sample_diffusion_vocaset.py:51, sample_diffusion_mead.py:67-86) -- 100-250 rows, where the out-proj / FFN2 GEMMs leave
# most CUs idle behind a 16- / 32-tile dependent k chain.  K slices (2 / 4) reduced by the LayerNorm launch that follows shorten the
# chain: +10...+19 % at 100-250 rows, -3...-5 % at 800 (profiles/r5_splitk/).  Results move by fp32 rounding (the k sum's association), so it
# is a property of the PLAN, chosen by the caller, never by the shape: a clip computes the same bits in every batch composition.
SINGLE_CLIP_PLAN = {"ksplit.out": 2, "ksplit.ffn2": 4}


class _FDMBase(ParamTree):
    preset_name = "vocaset"

    def _build(self, feature_dim, n_head, num_layers, struct, dtype, audio_encoder=True):
        base = presets.get(self.preset_name)
        self.preset = replace(base, name=f"{base.name}_d{feature_dim}", d=feature_dim, n_head=n_head, n_layers=num_layers,
                              ffn=2 * feature_dim, c=feature_dim // base.G)
        if self.preset.head_dim not in (64, 128, 256):
            raise FdmError(f"feature_dim/n_head = {self.preset.head_dim}: the HIP attention kernel supports head_dim 64/128/256")
        self.struct = struct
        presets.PRESETS[self.preset.name] = self.preset
        for k, v in synth.make_fdm_weights(self.preset.name).items():
            self._register(k, v)
        # nn.init.constant_(latent_decoder.*, 0)  (models/fdm_vocaset.py:50-51)
        self.latent_decoder.weight.data.zero_()
        self.latent_decoder.bias.data.zero_()
        n_pe = 630 if self.preset.pe == "periodic" else 5000
        self._register("PE.pe", schedule.positional_table(feature_dim, self.preset.pe, self.preset.period, n_pe).unsqueeze(0), buffer=True)
        # models/fdm.py:18-19 (wav2vec2-base-960h) vs models/fdm_vocaset.py:17 (hubert-large-ls960-ft)
        enc_cls, enc_path = (Wav2Vec2Model, "/data/WX/wav2vec2-base-960h") if self.preset_name == "biwi" else (HubertModel, "/data/WX/hubert-large-ls960-ft")
        if audio_encoder:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")       # the constructor's default path is the reference author's machine
                self.audio_encoder = enc_cls.from_pretrained(enc_path)
        else:
            self.audio_encoder = None
        self.one_hot_timesteps = None
        self._dtype = compute_dtype(dtype)
        self._plan = None
        self._plan_stale = True
        self._prep_key = None
        self._plan_options = {}                        # fdm_plan_set keys applied to every plan this module builds
        self._hub_key, self._hub = None, None          # _TensorKey of the encoded audio / "injected"

    # -- plan management --------------------------------------------------------------------
    def plan(self, device):
        if self._plan is None or self._plan_stale or self._plan.device != torch.device(device):
            sd = {k: v for k, v in self.state_dict().items() if not k.startswith("audio_encoder.")}
            self._plan = DenoiserPlan(self.preset, sd, self._dtype, device)
            for k, v in self._plan_options.items():
                self._plan.set(k, v)
            self._plan_stale = False
            self._prep_key = None
        return self._plan

    def set_plan_option(self, key, value):
        """fdm_plan_set(key, value) on this module's plan, now and after every rebuild (e.g. "ksplit.out" / "ksplit.ffn2": the
        single-clip setting, SINGLE_CLIP_PLAN below)."""
        self._plan_options[key] = int(value)
        if self._plan is not None and not self._plan_stale:
            self._plan.set(key, int(value))
            self._prep_key = None

    def set_audio_features(self, hub):
        """Inject precomputed HuBERT features [B, N, 1024] (bypasses the audio encoder)."""
        self._hub_key, self._hub = "injected", hub

    def audio_features(self, audio):
        if self._hub_key == "injected":
            return self._hub
        if not (isinstance(self._hub_key, _TensorKey) and self._hub_key.matches(audio)):
            # step-invariant: computed once per audio tensor OBJECT (hoisted, exact); a new tensor -- even one the
            # allocator placed at the same address -- is encoded again
            self._hub = self.audio_encoder(audio).last_hidden_state
            self._hub_key = _TensorKey(audio)
        return self._hub

    def prepare(self, audio, L, style, emo=None, cfg=False):
        """Per-batch tables of the plan.  style [B, n] (one condition per clip, the reference's call shape) or [B*S, n] with
        S > 1: S conditions per clip -- the style loop of samples/sample_diffusion_vocaset.py:71-83 as ONE step program
        (rows in (clip, condition) order; the audio encoder and the audio tables still run once per clip)."""
        hub = self.audio_features(audio)
        plan = self.plan(hub.device)
        key = (L, bool(cfg), tuple(style.flatten().tolist()), None if emo is None else tuple(emo.flatten().tolist()))
        pk = self._prep_key
        if not (pk is not None and pk[0].matches(hub) and pk[1] == key):
            B = hub.shape[0]
            st = style.reshape(-1, style.shape[-1])
            em = None if emo is None else emo.reshape(-1, emo.shape[-1])
            rows = max(st.shape[0], 1 if em is None else em.shape[0], B)
            if rows % B:
                raise FdmError(f"{rows} condition rows for {B} clips: expected a multiple (conditions per clip)")
            S = rows // B
            if st.shape[0] not in (1, rows) or (em is not None and em.shape[0] not in (1, rows)):
                raise FdmError("style / emotion one-hots must have one row, or one row per (clip, condition)")
            plan.prepare(hub, st if st.shape[0] == rows else st[0], em if (em is None or em.shape[0] == rows) else em[0],
                         L=L, cfg=cfg, n_conds=S)
            self._prep_key = (_TensorKey(hub), key)
        return plan

    def _forward(self, audio, t, vertice, style, emo=None):
        if not vertice.is_cuda:
            raise FdmError("FDM.forward runs on the HIP path only: move inputs to the GPU")
        if _mixed_t(t):      # one timestep per clip (the reference's [B] tensor): clips run one at a time, as its B = 1 calls do
            st = style.reshape(-1, style.shape[-1]) if torch.is_tensor(style) else style
            em = emo.reshape(-1, emo.shape[-1]) if torch.is_tensor(emo) else emo
            return _rows_one_by_one(self, lambda a, ti, v, s_, e_: self._forward(a, ti, v, s_, e_), t, vertice.shape[0], audio, vertice, st, em)
        G = self.preset.G
        L = vertice.shape[1] // G
        hub = self.audio_features(audio)
        nf = min(hub.shape[1] // self.preset.pair, L)                 # models/fdm_vocaset.py:64-66
        plan = self.prepare(audio, nf, style, emo)
        tt = _scalar_t(t)
        x = vertice.reshape(vertice.shape[0], L, G * vertice.shape[2])[:, :nf].reshape(vertice.shape[0], nf * G, -1)
        return plan.denoise(x.contiguous().float(), tt)


class FDM(_FDMBase):
    """VOCASET denoiser: FDM(feature_dim=512, n_head=8, num_layers=8, struct='Enc')."""
    preset_name = "vocaset"

    def __init__(self, feature_dim=512, n_head=8, num_layers=8, struct="Enc", dtype=None, audio_encoder=True):
        super().__init__()
        self._build(feature_dim, n_head, num_layers, struct, dtype, audio_encoder)

    def forward(self, audio, t, vertice, id_one_hot):
        return self._forward(audio, t, vertice, id_one_hot)


class FDMMead(_FDMBase):
    """3D-MEAD denoiser: FDM(feature_dim=512, vertice_dim=70110, n_head=4, num_layers=8, struct='Enc')."""
    preset_name = "mead"

    def __init__(self, feature_dim=512, vertice_dim=70110, n_head=4, num_layers=8, struct="Enc", dtype=None, audio_encoder=True):
        super().__init__()
        self._build(feature_dim, n_head, num_layers, struct, dtype, audio_encoder)

    def mask_cond(self, cond, train=False, force_mask=False):          # models/fdm_vqvae_mead.py:54-62
        if force_mask:
            return torch.zeros_like(cond)
        if train:
            mask = torch.bernoulli(torch.ones_like(cond) * 0.1)
            return cond * (1.0 - mask)
        return cond

    def forward(self, audio, t, vertice, emotion_one_hot, id_one_hot, mask_cond=False, train=True):
        emo = self.mask_cond(emotion_one_hot.reshape(-1, emotion_one_hot.shape[-1]), force_mask=bool(mask_cond))
        return self._forward(audio, t, vertice, id_one_hot, emo)


class FDMBiwi(_FDMBase):
    """BIWI denoiser (build-defined semantics: 'Dec' struct, latent regrouped x8; parity unpinned vs reference);
    audio encoder = wav2vec2-base (pinned against the reference, tests/golden/wav2vec.npz)."""
    preset_name = "biwi"

    def __init__(self, feature_dim=1024, vertice_dim=70110, n_head=4, num_layers=8, struct="Dec", dtype=None, audio_encoder=True):
        super().__init__()
        self._build(feature_dim, n_head, num_layers, struct, dtype, audio_encoder)

    def forward(self, audio, t, vertice, id_one_hot):
        return self._forward(audio, t, vertice, id_one_hot)


class ClassifierFreeSampleModel(nn.Module):
    """out_uncond + level * (out - out_uncond) (utiles/classifierfree.py:15-21), wired to the MEAD FDM:
    the unconditional pass zeroes the emotion one-hot; both passes run as one batched
    step program and the mix is fused into the scheduler kernel."""

    def __init__(self, model, level=2.5):
        super().__init__()
        self.model, self.level = model, level

    def forward(self, audio, t, x_noisy, emotion_one_hot, id_one_hot):
        m = self.model
        if _mixed_t(t):
            em = emotion_one_hot.reshape(-1, emotion_one_hot.shape[-1])
            st = id_one_hot.reshape(-1, id_one_hot.shape[-1])
            return _rows_one_by_one(m, lambda a, ti, x, e_, s_: self.forward(a, ti, x, e_, s_), t, x_noisy.shape[0], audio, x_noisy, em, st)
        L = x_noisy.shape[1] // m.preset.G
        plan = m.prepare(audio, L, id_one_hot, emotion_one_hot.reshape(-1, emotion_one_hot.shape[-1]), cfg=True)
        return plan.denoise(x_noisy.contiguous().float(), _scalar_t(t), cfg_scale=self.level)


# --------------------------------------------------------------------------------------------------
class GaussianDiffusion(nn.Module):
    """Sampling surface of the reference's GaussianDiffusion.  One class serves both reference variants:
    conditions are passed through (*cond) = (id_one_hot,) for VOCASET/BIWI, (emo_one_hot, id_one_hot) for MEAD.

    Build-added keyword inputs (default None => reference behaviour): noise= (x_T and per-step z for
    parity runs), seed=, t_range=, guidance_scale=."""

    def __init__(self, denoise_fn, *, text_use_bert_cls=False, channels=3, timesteps=1000, loss_type="l1",
                 use_dynamic_thres=False, dynamic_thres_percentile=0.9):
        super().__init__()
        self.channels, self.denoise_fn = channels, denoise_fn
        self.num_timesteps, self.loss_type = int(timesteps), loss_type
        self.text_use_bert_cls, self.use_dynamic_thres = text_use_bert_cls, use_dynamic_thres
        self.dynamic_thres_percentile = dynamic_thres_percentile
        for k, v in schedule.make_buffers(timesteps).items():
            self.register_buffer(k, v)
        self.full_chain = True     # False reproduces the VOCASET variant's hard-coded t = 999..500 (:663-665)

    # -- small exact host-side pieces kept for API completeness ---------------------------------
    @staticmethod
    def _extract(a, t, x_shape):
        out = a.gather(-1, t)
        return out.reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))

    def predict_noise_from_start(self, x_t, t, x0):
        return (self._extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - x0) / \
            self._extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape)

    def q_posterior(self, x_start, x_t, t):
        mean = self._extract(self.posterior_mean_coef1, t, x_t.shape) * x_start + \
            self._extract(self.posterior_mean_coef2, t, x_t.shape) * x_t
        return mean, self._extract(self.posterior_variance, t, x_t.shape), \
            self._extract(self.posterior_log_variance_clipped, t, x_t.shape)

    def q_sample(self, x_start, t, noise=None):
        """sqrt(abar_t) x0 + sqrt(1 - abar_t) eps (:729-735) through the fused scheduler kernel (all clips share t)."""
        if not x_start.is_cuda:
            raise FdmError("GaussianDiffusion.q_sample runs on the HIP path only: move the latents to the GPU")
        noise = torch.randn_like(x_start) if noise is None else noise
        if _mixed_t(t):      # per-clip timesteps (p_losses draws them: :759): the update is elementwise, one clip per launch
            tf = t.flatten()
            return torch.cat([self.q_sample(x_start[i:i + 1], tf[i:i + 1], noise[i:i + 1]) for i in range(x_start.shape[0])])
        from . import ops
        x0 = x_start.float().contiguous()
        z = noise.float().contiguous()
        out = torch.empty_like(x0)
        zero = torch.zeros(self.num_timesteps, device=x0.device)
        tt = torch.tensor([_scalar_t(t)], dtype=torch.int32, device=x0.device)
        ops.sched_step(0, x0, z, out, x0.numel(), tseq=tt, c1=self.sqrt_alphas_cumprod, c2=self.sqrt_one_minus_alphas_cumprod,
                       sigma=zero, noise=z)
        return out

    def p_losses(self, x_start, t, audio, *cond, noise=None):
        """Forward value of the training loss (:737-755): q_sample -> denoiser -> mean |x0 - x0_hat|^p."""
        from . import ops
        x_noisy = self.q_sample(x_start, t, noise)
        x_recon = self.denoise_fn(audio, t, x_noisy, *cond)
        xs = x_start[:, : x_recon.shape[1]].float().contiguous()
        return ops.mean_diff(xs, x_recon.contiguous(), l1=(self.loss_type == "l1"))[0], x_recon

    def _split_cond(self, cond):
        m = self.denoise_fn
        model = m.model if isinstance(m, ClassifierFreeSampleModel) else m
        if model.preset.n_emo:
            emo, style = cond
            return model, style, emo.reshape(-1, emo.shape[-1])
        return model, cond[0], None

    def _plan(self, audio, shape, cond, guidance_scale=None):
        model, style, emo = self._split_cond(cond)
        cfg = isinstance(self.denoise_fn, ClassifierFreeSampleModel) or guidance_scale is not None
        scale = guidance_scale if guidance_scale is not None else getattr(self.denoise_fn, "level", 2.5)
        L = shape[1] // model.preset.G
        return model.prepare(audio, L, style, emo, cfg=cfg), scale

    @torch.no_grad()
    def p_mean_variance(self, x, t, clip_denoised, audio, *cond):
        x_recon = self.denoise_fn(audio, t, x, *cond)
        return self.q_posterior(x_start=x_recon, x_t=x, t=t)

    @torch.no_grad()
    def p_sample(self, x, t, audio, *cond, clip_denoised=False, noise=None):
        """One reverse step on the HIP path (denoiser + fused scheduler update)."""
        if _mixed_t(t):
            model = self._split_cond(cond)[0]
            z = noise if noise is not None else torch.randn_like(x)
            rows = x.shape[0]
            cr = [c.reshape(-1, c.shape[-1]) if torch.is_tensor(c) else c for c in cond]
            return _rows_one_by_one(model, lambda a, ti, xi, zi, *ci: self.p_sample(xi, ti, a, *ci, noise=zi), t, rows, audio, x, z, *cr)
        plan, scale = self._plan(audio, x.shape, cond)
        tt = _scalar_t(t)
        z = noise if noise is not None else torch.randn_like(x)
        return plan.sample_ddpm(x.float().contiguous(), [tt], noise=z.reshape(1, *x.shape), cfg_scale=scale, use_graph=False)

    @torch.no_grad()
    def p_sample_loop(self, shape, audio, *cond, noise=None, seed=None, t_range=None, guidance_scale=None, x_T=None, clip0=0):
        """Build-added keywords (default = reference behaviour): noise=(x_T, z[T]) injects everything (parity runs); x_T= fixes the
        start with in-kernel Philox noise keyed by (seed, clip0 + row block, step): a clip's result depends on its index, not on
        the batch it is sampled in."""
        plan, scale = self._plan(audio, shape, cond, guidance_scale)
        dev = plan.device
        if t_range is None:
            t_range = (self.num_timesteps - 1, -1) if self.full_chain else (999, 499)
        ts = list(range(t_range[0], t_range[1], -1))
        if noise is not None:
            x_T, z = noise[0].to(dev), noise[1]
            return plan.sample_ddpm(x_T.float().contiguous(), ts, noise=z, cfg_scale=scale)
        x_T = torch.randn((plan.B,) + tuple(shape[1:]), device=dev) if x_T is None else x_T.to(dev).float().contiguous()
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if seed is None else int(seed)
        return plan.sample_ddpm(x_T, ts, seed=seed, clip0=int(clip0), cfg_scale=scale)

    @torch.no_grad()
    def sample(self, audio, latent_motion_shape, *cond, **kw):
        return self.p_sample_loop(latent_motion_shape, audio, *cond, **kw)

    @torch.no_grad()
    def ddim_sample(self, audio, latent_motion_shape, id_one_hot, steps=500, *, x_T=None, guidance_scale=None):
        """id_one_hot [1, n] / [B, n]: the reference's call.  [B*S, n] with S > 1: the S style conditions of every clip in
        one call (the sampler's style loop, samples/sample_diffusion_vocaset.py:71-83, batched); returns [B*S, L*G, c] in
        (clip, condition) order, each block bit-identical to the one-condition call with the same x_T block."""
        plan, scale = self._plan(audio, latent_motion_shape, (id_one_hot,), guidance_scale)
        shape = (plan.B,) + tuple(latent_motion_shape[1:])
        x_T = torch.randn(shape, device=plan.device) if x_T is None else x_T.to(plan.device)
        return plan.sample_ddim(x_T.float().contiguous(), steps, cfg_scale=scale)

    def forward(self, x, audio, *cond):
        """Forward-only loss (:757-761); there is no backward pass on this path (training is out of scope).  Like the reference,
        one timestep per clip: t = randint(0, T, (b,)) (:759).  The step program shares t across its rows, so clips with
        different t run as separate B = 1 calls (the reference itself can only run B = 1); equal-sized clips make the mean of
        the per-clip losses the batch loss."""
        b = x.shape[0]
        t = torch.randint(0, self.num_timesteps, (b,), device=x.device).long()
        if b == 1 or bool((t == t[0]).all()):
            return self.p_losses(x, t, audio, *cond)
        m = self.denoise_fn.model if isinstance(self.denoise_fn, ClassifierFreeSampleModel) else self.denoise_fn
        inj = m._hub if m._hub_key == "injected" else None          # precomputed audio features [B, N, fw]: one clip at a time too
        losses, recons = [], []
        try:
            for i in range(b):
                if inj is not None:
                    m.set_audio_features(inj[i:i + 1])
                ci = tuple(c[i:i + 1] if (torch.is_tensor(c) and c.dim() > 1 and c.shape[0] == b) else c for c in cond)
                li, ri = self.p_losses(x[i:i + 1], t[i:i + 1], audio[i:i + 1], *ci)
                losses.append(li)
                recons.append(ri)
        finally:
            if inj is not None:
                m.set_audio_features(inj)
        return torch.stack(losses).mean(), torch.cat(recons)


# --------------------------------------------------------------------------------------------------
class VQAutoEncoder(ParamTree):
    """quant + decode of the (E)VQ-VAE; `args` as returned by models/utils/config.py helpers."""

    def __init__(self, args, dtype=None):
        super().__init__()
        self.args = args
        if args.in_dim == 70110:
            base = presets.BIWI
        elif args.n_embed > 256:
            base = presets.MEAD
        else:
            base = presets.VOCASET
        self.preset = replace(base, name=f"vq_{base.name}_{args.in_dim}_{args.face_quan_num}", G=args.face_quan_num,
                              c=args.zquant_dim, V3=args.in_dim, n_books=max(1, args.n_embed // 256))
        presets.PRESETS[self.preset.name] = self.preset
        for k, v in synth.make_vq_weights(self.preset.name, encoder=True).items():
            self._register(k, v)
        self._register("encoder.encoder_pos_embedding.pe",
                       schedule.positional_table(presets.VQ_HIDDEN, "sinus", 1, 5000).unsqueeze(1), buffer=True)
        self._register("decoder.decoder_pos_embedding.pe",
                       schedule.positional_table(presets.VQ_HIDDEN, "sinus", 1, 5000).unsqueeze(1), buffer=True)
        self._dtype = compute_dtype(dtype)
        self._plan = None
        self._plan_stale = True

    def plan(self, device):
        if self._plan is None or self._plan_stale or self._plan.device != torch.device(device):
            self._plan = VQPlan(self.preset, self.state_dict(), _side_dtype(self._dtype), device)
            self._plan_stale = False
        return self._plan

    def encode(self, x, one_hot=None):
        """x [B, L, V3] (vertices minus template) -> latent [B, L*G, c] (models/vq_vae_vocaset.py:23-28)."""
        if not x.is_cuda:
            raise FdmError("VQAutoEncoder.encode runs on the HIP path only")
        emo = None if one_hot is None else one_hot.reshape(-1, one_hot.shape[-1])
        return self.plan(x.device).encode(x, emo)

    def quant(self, x, one_hot=None, stats=True):
        """-> (z_q [B, c, L*G], emb_loss, (perplexity, min_encodings [B*L*G, 256], indices [B*L*G, 1])): the reference's
        whole tuple (models/vq_vae_vocaset.py:31-33 -> models/lib/quantizer.py:35-64, beta = 0.25), all of it on the device.
        stats=False (what the sampling pipeline passes: it uses z_q only, as every sampler of the reference does) skips the
        training-side outputs -- no [B*L*G, 256] one-hot, no statistics kernels: (z_q, None, (None, None, indices))."""
        if not x.is_cuda:
            raise FdmError("VQAutoEncoder.quant runs on the HIP path only")
        emo = None if one_hot is None else one_hot.reshape(-1, one_hot.shape[-1])
        if not stats:
            zq, idx = self.plan(x.device).quant(x, emo)
            return zq, None, (None, None, idx)
        return self.plan(x.device).quant_full(x, emo, beta=0.25)

    def decode(self, quant):
        if not quant.is_cuda:
            raise FdmError("VQAutoEncoder.decode runs on the HIP path only")
        return self.plan(quant.device).decode(quant)
