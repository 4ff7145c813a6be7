"""Denoiser + scheduler plan: a thin binding of the library's plan layer (include/fdm_hip.h: fdm_plan_create,
fdm_plan_set_weights, fdm_audio_prepare, fdm_denoise_step, fdm_sample_graph; implementation csrc/plan.hip).

The per-step FDM forward (models/fdm_vocaset.py:54-91, models/fdm_vqvae_mead.py:65-104) and the sampling loops
(diffusion_BIWI_encoder_decoder.py:649-710, diffusion_mead_encoder_decoder.py:649-667) live in C++ / HIP:
  * commit (once per model): operand-kind weight copies; tau table = Mish(W_t[:, t] + b_t) for all 1000 t (the one-hot
    GEMV of :71-72 is a column gather); per-layer time tables TT_l[t] = Wo_l (Wv_l tau_t) of the folded cross-attention.
  * prepare (once per batch of clips): AF = audio_extract(audio-encoder features); per-layer tables
    C1_l = Wo_l (Wv_l AF + bv_l) + bo_l; E0 = PE[l] + style[b] (+ emotion[b]).  The cross-attention's memory mask leaves
    exactly one key per query (models/fdm_vocaset.py:119-127), so CA_l(h, AF + tau)[i] = C1_l[i] + TT_l[t] exactly.
  * step program (recorded once, captured into a hipGraph, `graph_steps` diffusion steps per graph launch; t comes from a
    device counter): latent_encoder GEMM(+bias+Mish+E0) -> n_layers x { QKV GEMM (K / V written fragment-packed) -> fused
    causal-ALiBi attention -> out-proj GEMM(+bias+residual) -> fused LN1+LN2(+C1_l + TT_l[t]) -> FFN1 GEMM(+bias+ReLU) ->
    FFN2 GEMM(+bias+residual) -> LN3 } -> latent_decoder GEMM with the scheduler update (DDPM / DDIM, Philox or injected
    noise) in its epilogue; CFG plans run cond + uncond rows through the same launches and mix in the scheduler kernel.
This module only moves pointers: torch owns the caller-side tensors and the stream."""
import ctypes as C

import torch

from . import presets, schedule
from ._lib import F32, FdmError, ModelDesc, SampleArgs, check, lib

TILE_SITES = ("enc", "qkv", "qkv_ln", "out", "out_ln", "ffn1", "ffn2", "ffn2_stat", "dec", "dec_ln")


def _dev(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def model_desc(p):
    """fdm_model_desc of a presets.Preset (the reference constructors' numbers)."""
    return ModelDesc(p.d, p.n_head, p.n_layers, p.ffn, p.G, p.c, p.n_style, p.n_emo, p.audio_in, p.pair,
                     1 if p.pe == "periodic" else 0, p.period, int(p.latent_mish), int(p.style_mish), p.max_len)


class DenoiserPlan:
    def __init__(self, preset, weights, dtype=F32, device="cuda:0"):
        self.p = presets.get(preset)
        self.dtype = dtype
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise FdmError("DenoiserPlan runs on the HIP path only (no CPU fallback)")
        self.h = None
        self.B = self.L = self.M = 0
        self.S = 1
        self.cfg = False
        self._keep, self._inputs = [], None
        h = C.c_void_p()
        desc = model_desc(self.p)
        with torch.cuda.device(self.device):
            check(lib().fdm_plan_create(C.byref(desc), 1, 1, 0, dtype, C.byref(h)))
            self.h = h
            # schedule tables and the positional table as the reference computes them (torch fp64 -> fp32, torch fp32
            # sin / cos / exp), so that sampling is bit-identical to the reference's expressions; C-only callers get the
            # library's own (fdm_schedule_host, fdm_pe_table_host: within 1 ulp of these)
            buf = schedule.make_buffers(1000)
            c1, c2, sg = schedule.ddpm_tables(buf)
            tabs = {"sched.c1": c1, "sched.c2": c2, "sched.sigma": sg, "sched.sra": buf["sqrt_recip_alphas_cumprod"],
                    "sched.srm1": buf["sqrt_recipm1_alphas_cumprod"],
                    "PE.pe": schedule.positional_table(self.p.d, self.p.pe, self.p.period, self.p.max_len + 30)}
            keep = []
            for k, v in list(weights.items()) + list(tabs.items()):
                if k.startswith("audio_encoder.") or (k == "PE.pe" and v is not tabs["PE.pe"]):
                    continue        # (the PE buffer is rebuilt from the constructor's numbers, as the reference's __init__ does)
                t = v.detach().to(torch.float32).contiguous()       # host or device memory: the plan copies it
                keep.append(t)
                check(lib().fdm_plan_set_weights(h, k.encode(), t.data_ptr(), t.numel(), _stream()))
            torch.cuda.current_stream().synchronize()               # the copies read `keep`
            check(lib().fdm_plan_commit(h, _stream()))

    def __del__(self):
        try:
            if self.h:
                lib().fdm_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    def prepare(self, hub, style, emo=None, L=None, cfg=False, n_conds=1):
        """hub [B, N, fw] audio-encoder features; style [B, n_style]; emo [B, n_emo]; L latent frames.

        n_conds = S > 1: S conditions per clip in ONE step program (fdm_audio_prepare_conds) -- what the reference's
        samplers do as S sequential B = 1 calls with the same audio (samples/sample_diffusion_vocaset.py:71-83).  style
        [B*S, n_style] / emo [B*S, n_emo] in (clip, condition) order; the plan's batch becomes B*S row blocks (x_T, noise,
        outputs are [B*S, L*G, c]); the audio tables are computed once per clip and shared by its conditions."""
        p, dv = self.p, self.device
        hub = _dev(hub, dv)
        B, N, fw = hub.shape
        S = int(n_conds)
        if S < 1:
            raise FdmError(f"n_conds={S}")
        nfa = N // p.pair
        L = nfa if L is None else min(L, nfa)
        if L < 1 or L > p.max_len:
            raise FdmError(f"latent frames L={L} outside [1, {p.max_len}] (models/fdm_vocaset.py:44)")
        if style.dim() == 1:
            style = style.unsqueeze(0).expand(B * S, -1)
        if style.shape[0] != B * S:
            raise FdmError(f"style has {style.shape[0]} rows, expected clips x conditions = {B * S}")
        style = _dev(style, dv)
        if p.n_emo:
            if emo is None:
                raise FdmError("this preset needs an emotion one-hot")
            if emo.dim() == 1:
                emo = emo.unsqueeze(0).expand(B * S, -1)
            if emo.shape[0] != B * S:
                raise FdmError(f"emotion one-hot has {emo.shape[0]} rows, expected clips x conditions = {B * S}")
            emo = _dev(emo, dv)
        with torch.cuda.device(dv):
            check(lib().fdm_audio_prepare_conds(self.h, hub.data_ptr(), B, N, fw, S, style.data_ptr(),
                                                emo.data_ptr() if p.n_emo else None, L, int(bool(cfg)), _stream()))
        # the library reads hub / style / emo asynchronously on this stream: keep them alive until the next prepare
        self._inputs = (hub, style, emo)
        self.B, self.L, self.M, self.cfg, self.S = B * S, L, B * S * L, bool(cfg), S
        return L

    def _check_x(self, x):
        if self.M == 0:
            raise FdmError("call prepare() first")
        if tuple(x.shape) != (self.B, self.L * self.p.G, self.p.c):
            raise FdmError(f"latent shape {tuple(x.shape)} != {(self.B, self.L * self.p.G, self.p.c)}")
        return _dev(x, self.device)

    # ------------------------------------------------------------------------------------------
    def denoise(self, x, t, cfg_scale=2.5, return_uncond=False):
        """One FDM.forward: x [B, L*G, c] -> x0_hat [B, L*G, c] (CFG-mixed when prepared with cfg=True)."""
        x = self._check_x(x)
        out = torch.empty_like(x)
        unc = torch.empty_like(x) if (return_uncond and self.cfg) else None
        with torch.cuda.device(self.device):
            check(lib().fdm_denoise_step(self.h, x.data_ptr(), int(t), float(cfg_scale), out.data_ptr(),
                                         unc.data_ptr() if unc is not None else None, _stream()))
        return (out, unc) if return_uncond else out

    def _sample(self, a, x_T):
        x = self._check_x(x_T)
        out = torch.empty_like(x)
        a.x_T, a.out = x.data_ptr(), out.data_ptr()
        with torch.cuda.device(self.device):
            check(lib().fdm_sample_graph(self.h, C.byref(a), _stream()))
        return out

    def sample_ddpm(self, x_T, t_list, noise=None, seed=0, clip0=0, cfg_scale=2.5, use_graph=True, record=None, graph_steps=0):
        """p_sample_loop over t_list (descending).  noise [len(t_list), B, L*G, c] injects z per step; otherwise z is
        drawn in-kernel (Philox keyed by seed, global clip index clip0 + b, step).  record (a list) receives the latent
        after every step."""
        a = SampleArgs()
        ts = (C.c_int * len(t_list))(*[int(t) for t in t_list])
        a.kind, a.t_list, a.n_steps = 0, C.cast(ts, C.c_void_p), len(t_list)
        a.seed, a.clip0, a.cfg_scale, a.eager, a.graph_steps = int(seed), int(clip0), float(cfg_scale), int(not use_graph), int(graph_steps)
        if noise is not None:
            nz = _dev(noise, self.device)
            if nz.numel() != len(t_list) * self.M * self.p.d:
                raise FdmError("noise must be [len(t_list), B, L*G, c]")
            self._keep = (self._keep + [nz])[-8:]          # recorded programs (at most 8) point at it
            a.noise = nz.data_ptr()
        rec = None
        if record is not None:
            rec = torch.empty(len(t_list), *x_T.shape, device=self.device)
            a.record = rec.data_ptr()
        out = self._sample(a, x_T)
        if rec is not None:
            record.extend(rec[i] for i in range(len(t_list)))
        return out

    def sample_ddim(self, x_T, steps, cfg_scale=2.5, use_graph=True, graph_steps=0, record=None):
        """ddim_sample (eta = 0).  The last pair (t, -1) never updates the latent in the reference (:695-696), so its
        denoiser call is skipped: exact.  record (a list) receives the latent after every live pair."""
        a = SampleArgs()
        a.kind, a.ddim_steps, a.cfg_scale, a.eager, a.graph_steps = 1, int(steps), float(cfg_scale), int(not use_graph), int(graph_steps)
        rec = None
        if record is not None:
            n_live = sum(1 for pr in schedule.ddim_time_pairs(int(steps)) if pr[1] >= 0)
            rec = torch.empty(max(n_live, 1), *x_T.shape, device=self.device)
            a.record = rec.data_ptr()
        out = self._sample(a, x_T)
        if rec is not None:
            record.extend(rec[i] for i in range(n_live))
        return out

    # ------------------------------------------------------------------------------------------
    def tune(self):
        """Tune the GEMM tiles for the prepared shape now (plan-time work).  Nothing else tunes: get("needs_tune") says when a
        shape has run 2000 steps untuned; only a caller that set "tune_lazy" lets a sampling call tune (once per shape)."""
        with torch.cuda.device(self.device):
            check(lib().fdm_plan_tune(self.h, _stream()))

    def get(self, key):
        v = C.c_longlong()
        check(lib().fdm_plan_get(self.h, key.encode(), C.byref(v)))
        return v.value

    def set(self, key, value):
        check(lib().fdm_plan_set(self.h, key.encode(), int(value)))

    @property
    def tiles(self):
        """Tile chosen per GEMM call site of the step (0 = library heuristic)."""
        return {k: self.get("tile." + k) for k in TILE_SITES}

    @property
    def fuse_ln3(self):
        return bool(self.get("fuse_ln3"))
