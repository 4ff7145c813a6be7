"""Denoiser + scheduler plan: the per-step FDM forward (models/fdm_vocaset.py:54-91,
models/fdm_vqvae_mead.py:65-104) and the sampling loops (diffusion_BIWI_encoder_decoder.py:649-710,
diffusion_mead_encoder_decoder.py:649-667) as one recorded HIP step program.

What runs where
  * init (once per model): operand-dtype weight copies; tau table = Mish(W_t[:, t] + b_t) for all
    1000 t (the one-hot GEMV of :71-72 is a column gather); per-layer time tables
    TT_l[t] = Wo_l (Wv_l tau_t) of the cross-attention.
  * prepare (once per batch of clips): AF = audio_extract(HuBERT features); per-layer tables
    C1_l = Wo_l (Wv_l AF + bv_l) + bo_l; E0 = PE[l] + style[b] (+ emotion[b]).
    The cross-attention's memory mask leaves exactly one key per query (models/fdm_vocaset.py:119-127), so
    CA_l(h, AF + tau)[i] = C1_l[i] + TT_l[t] exactly (softmax over one key == 1).
  * step program (captured into one hipGraph, replayed T times; t comes from a device counter):
      [cast x] -> latent_encoder GEMM(+bias+Mish+E0) -> 8 x { QKV GEMM (V scattered transposed) ->
      fused causal-ALiBi attention -> out-proj GEMM(+bias+residual) -> LN1 -> LN2(+C1_l + TT_l[t]) ->
      FFN1 GEMM(+bias+ReLU) -> FFN2 GEMM(+bias+residual) -> LN3 } -> latent_decoder GEMM ->
      fused scheduler update (DDPM / DDIM, optional CFG mix, Philox or injected noise).
torch only owns the device buffers and the stream."""
import math

import torch

from . import ops, presets, schedule
from ._lib import ACT_MISH, ACT_NONE, ACT_RELU, BF16, F32, FdmError


def _dev(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class DenoiserPlan:
    def __init__(self, preset, weights, dtype=F32, device="cuda:0"):
        self.p = presets.get(preset)
        self.dtype = dtype
        self.device = torch.device(device)
        self.td = ops.tdtype(dtype)
        p, dv = self.p, self.device
        if p.head_dim != 128:
            raise FdmError("denoiser head_dim must be 128")
        self.stream = torch.cuda.Stream(device=dv)
        w = {k: _dev(v, dv) for k, v in weights.items() if not k.startswith("audio_encoder.") and k != "PE.pe"}
        self.w32 = w
        need = ["audio_extract.0.weight", "audio_extract.2.weight", "time_embedd.0.weight", "style_embedd.weight",
                "latent_encoder.0.weight", "latent_decoder.weight"]
        for k in need:
            if k not in w:
                raise FdmError(f"missing weight {k}")
        d = p.d
        # operand-dtype copies of the per-step matrices
        self.wt = {}
        step_mats = ["latent_encoder.0.weight", "latent_decoder.weight"]
        for l in range(p.n_layers):
            pre = f"transformer_decoder.layers.{l}."
            step_mats += [pre + "self_attn.in_proj_weight", pre + "self_attn.out_proj.weight",
                          pre + "linear1.weight", pre + "linear2.weight"]
        with torch.cuda.stream(self.stream):
            for k in step_mats:
                self.wt[k] = ops.to_operand(w[k], dtype)
            # tau table [1000, d] = Mish(W_t^T + b_t)   (models/fdm_vocaset.py:71-72)
            wtT = w["time_embedd.0.weight"].t().contiguous()
            self.tau = torch.empty(1000, d, device=dv)
            ops.bias_act(wtT, w["time_embedd.0.bias"], self.tau, 1000, d, ACT_MISH)
            # per-layer time tables TT_l = (tau Wv^T) Wo^T   (fp32 MFMA, one-time)
            self.TT, self.Wv, self.bv, self.Wo, self.bo = [], [], [], [], []
            tmp = torch.empty(1000, d, device=dv)
            for l in range(p.n_layers):
                pre = f"transformer_decoder.layers.{l}.multihead_attn."
                Wv = w[pre + "in_proj_weight"][2 * d:].contiguous()
                bv = w[pre + "in_proj_bias"][2 * d:].contiguous()
                Wo, bo = w[pre + "out_proj.weight"], w[pre + "out_proj.bias"]
                tt = torch.empty(1000, d, device=dv)
                ops.gemm(self.tau, Wv, 1000, d, d, out_f32=tmp)
                ops.gemm(tmp, Wo, 1000, d, d, out_f32=tt)
                self.TT.append(tt)
                self.Wv.append(Wv); self.bv.append(bv); self.Wo.append(Wo); self.bo.append(bo)
            self.slopes = torch.tensor(schedule.alibi_slopes(p.n_head), dtype=torch.float32).to(dv)
            self.pe = schedule.positional_table(d, p.pe, p.period, p.max_len + 30).to(dv)
        self.stream.synchronize()
        self.buf = {k: v.to(dv) for k, v in schedule.make_buffers(1000).items()}
        c1, c2, sg = schedule.ddpm_tables(schedule.make_buffers(1000))
        self.c1, self.c2, self.sigma = c1.to(dv), c2.to(dv), sg.to(dv)
        self.B = self.L = self.M = 0
        self._progs = {}
        self._ddim = {}

    # ------------------------------------------------------------------------------------------
    def prepare(self, hub, style, emo=None, L=None, cfg=False):
        """hub [B, N, 1024] HuBERT features; style [B, n_style]; emo [B, n_emo]; L latent frames."""
        p, dv, d = self.p, self.device, self.p.d
        hub = _dev(hub, dv)
        B, N = hub.shape[0], hub.shape[1]
        nfa = N // p.pair
        L = nfa if L is None else min(L, nfa)
        if L < 1 or L > p.max_len:
            raise FdmError(f"latent frames L={L} outside [1, {p.max_len}] (models/fdm_vocaset.py:44)")
        if style.dim() == 1:
            style = style.unsqueeze(0).expand(B, -1)
        style = _dev(style, dv)
        if p.n_emo:
            if emo is None:
                raise FdmError("this preset needs an emotion one-hot")
            if emo.dim() == 1:
                emo = emo.unsqueeze(0).expand(B, -1)
            emo = _dev(emo, dv)
        M = B * L
        self.B, self.L, self.M, self.cfg = B, L, M, bool(cfg)
        self._progs = {}
        w = self.w32
        with torch.cuda.stream(self.stream):
            # audio rows: pair HuBERT frames (models/fdm_vqvae_mead.py:73), crop to L (:64-66)
            a = hub[:, : nfa * p.pair].reshape(B, nfa, p.pair * 1024)[:, :L].reshape(M, p.pair * 1024).contiguous()
            t1 = torch.empty(M, d, device=dv)
            AF = torch.empty(M, d, device=dv)
            ops.gemm(a, w["audio_extract.0.weight"], M, d, p.audio_in, bias=w["audio_extract.0.bias"], act=ACT_MISH, out_f32=t1)
            ops.gemm(t1, w["audio_extract.2.weight"], M, d, d, bias=w["audio_extract.2.bias"], out_f32=AF)
            self.AF = AF
            # folded cross-attention tables C1_l [M, d]
            self.C1 = []
            for l in range(p.n_layers):
                c1 = torch.empty(M, d, device=dv)
                ops.gemm(AF, self.Wv[l], M, d, d, bias=self.bv[l], out_f32=t1)
                ops.gemm(t1, self.Wo[l], M, d, d, bias=self.bo[l], out_f32=c1)
                self.C1.append(c1)
            # conditioning addend E0 = PE[l] + style[b] (+ emotion[b])  (:75-84)
            sty = torch.empty(B, d, device=dv)
            ops.small_linear(style, w["style_embedd.weight"], w["style_embedd.bias"], sty, B, p.n_style, d,
                             ACT_MISH if p.style_mish else ACT_NONE)
            rows = (2 * M) if cfg else M
            self.E0 = torch.empty(rows, d, device=dv)
            if p.n_emo:
                em = torch.empty(B, d, device=dv)
                ops.small_linear(emo, w["emotion_embedd.weight"], w["emotion_embedd.bias"], em, B, p.n_emo, d)
                ops.add_rows(self.E0[:M], M, d, self.pe, 1, L, sty, L, B, em, L, B)
                if cfg:   # null condition = zeros_like(emotion one-hot) (models/fdm_vqvae_mead.py:56-57) -> bias only
                    emu = torch.empty(B, d, device=dv)
                    ops.small_linear(torch.zeros_like(emo), w["emotion_embedd.weight"], w["emotion_embedd.bias"], emu, B, p.n_emo, d)
                    ops.add_rows(self.E0[M:], M, d, self.pe, 1, L, sty, L, B, emu, L, B)
            else:
                ops.add_rows(self.E0[:M], M, d, self.pe, 1, L, sty, L, B)
                if cfg:
                    ops.add_rows(self.E0[M:], M, d, self.pe, 1, L, sty, L, B)
            if cfg:
                self.C1 = [torch.cat([c, c]) for c in self.C1]
            self._alloc_workspace()
        self.stream.synchronize()
        return L

    def _alloc_workspace(self):
        p, dv, d = self.p, self.device, self.p.d
        R = (2 * self.M) if self.cfg else self.M          # rows through the decoder stack
        BB = (2 * self.B) if self.cfg else self.B
        td = self.td
        z = lambda *s, dt=torch.float32: torch.zeros(*s, device=dv, dtype=dt)
        self.R, self.BB = R, BB
        self.Lpad = (self.L + 31) // 32 * 32
        ws = dict(h=z(R, d), h2=z(R, d), x1=z(R, d), x0=z(R, d), x=z(self.M, d))
        if self.dtype == BF16:
            ws.update(xt=z(self.M, d, dt=td), ht=z(R, d, dt=td), h2t=z(R, d, dt=td))
        else:   # fp32 operands alias the fp32 residual-stream buffers
            ws.update(xt=ws["x"], ht=ws["h"], h2t=ws["h2"])
        ws.update(qkv=z(R, 3 * d, dt=td), ctx=z(R, d, dt=td), u=z(R, p.ffn, dt=td),
                  vt=z(BB * p.n_head, p.head_dim, self.Lpad, dt=td))
        self.ws = ws
        self.step = torch.zeros(1, dtype=torch.int32, device=dv)
        self.tseq = torch.zeros(1024, dtype=torch.int32, device=dv)

    # ------------------------------------------------------------------------------------------
    def _record_pass(self):
        """Record one denoiser pass: ws['x'] (fp32 [M, d]) -> ws['x0'] ([R, d], cond rows then uncond rows)."""
        p, d, M, R, ws, w, wt = self.p, self.p.d, self.M, self.R, self.ws, self.w32, self.wt
        both = self.dtype == BF16
        if both:
            ops.cast(ws["x"], ws["xt"])
        for half in range(2 if self.cfg else 1):
            o = half * M
            ops.gemm(ws["xt"], wt["latent_encoder.0.weight"], M, d, d, bias=w["latent_encoder.0.bias"],
                     act=ACT_MISH if p.latent_mish else ACT_NONE, resid=self.E0[o:], out_f32=ws["h"][o:],
                     out_t=ws["ht"][o:] if both else None)
        for l in range(p.n_layers):
            pre = f"transformer_decoder.layers.{l}."
            ops.gemm(ws["ht"], wt[pre + "self_attn.in_proj_weight"], R, 3 * d, d, bias=w[pre + "self_attn.in_proj_bias"],
                     out_t=ws["qkv"], ldo_t=3 * d, out_vt=ws["vt"], vt_col0=2 * d, vt_L=self.L, vt_Lpad=self.Lpad,
                     vt_hd=p.head_dim)
            ops.attention(ws["qkv"], ws["qkv"][:, d:], ws["vt"], ws["ctx"], B=self.BB, H=p.n_head, L=self.L, hd=p.head_dim,
                          ldq=3 * d, ldk=3 * d, ldo=d, Lpad=self.Lpad, scale=1.0 / math.sqrt(p.head_dim), causal=True,
                          slopes=self.slopes, period=p.period)
            ops.gemm(ws["ctx"], wt[pre + "self_attn.out_proj.weight"], R, d, d, bias=w[pre + "self_attn.out_proj.bias"],
                     resid=ws["h"], out_f32=ws["x1"])
            ops.layernorm(ws["x1"], w[pre + "norm1.weight"], w[pre + "norm1.bias"], R, d, y_f32=ws["h"])
            ops.layernorm(ws["h"], w[pre + "norm2.weight"], w[pre + "norm2.bias"], R, d, add_mat=self.C1[l],
                          add_tab=self.TT[l], tab_index=self.tseq, tab_step=self.step, y_f32=ws["h2"],
                          y_t=ws["h2t"] if both else None, dtype=self.dtype)
            ops.gemm(ws["h2t"], wt[pre + "linear1.weight"], R, p.ffn, d, bias=w[pre + "linear1.bias"], act=ACT_RELU,
                     out_t=ws["u"])
            ops.gemm(ws["u"], wt[pre + "linear2.weight"], R, d, p.ffn, bias=w[pre + "linear2.bias"], resid=ws["h2"],
                     out_f32=ws["x1"])
            ops.layernorm(ws["x1"], w[pre + "norm3.weight"], w[pre + "norm3.bias"], R, d, y_f32=ws["h"],
                          y_t=ws["ht"] if both else None, dtype=self.dtype)
        ops.gemm(ws["ht"], wt["latent_decoder.weight"], R, d, d, bias=w["latent_decoder.bias"], out_f32=ws["x0"])

    def _program(self, kind, **kw):
        """Build (once) the step program `kind` in {'pass', 'ddpm', 'ddim'}."""
        key = (kind,) + tuple(sorted((k, (v.data_ptr() if torch.is_tensor(v) else v)) for k, v in kw.items()))
        if key in self._progs:
            return self._progs[key]
        n = self.M * self.p.d
        ws = self.ws
        prog = ops.Program()
        with prog:
            self._record_pass()
            x0u = ws["x0"][self.M:] if self.cfg else None
            if kind == "ddpm":
                ops.sched_step(0, ws["x0"], ws["x"], ws["x"], n, x0u=x0u, cfg_scale=kw.get("cfg_scale", 0.0),
                               n_per_clip=self.L * self.p.d, tseq=self.tseq, step=self.step, advance=1,
                               c1=self.c1, c2=self.c2, sigma=self.sigma, noise=kw.get("noise"), seed=kw.get("seed", 0),
                               clip0=kw.get("clip0", 0))
            elif kind == "ddim":
                ops.sched_step(1, ws["x0"], ws["x"], ws["x"], n, x0u=x0u, cfg_scale=kw.get("cfg_scale", 0.0),
                               tseq=self.tseq, step=self.step, advance=1, sra=self.buf["sqrt_recip_alphas_cumprod"],
                               srm1=self.buf["sqrt_recipm1_alphas_cumprod"], sqrt_an=kw["sqrt_an"], c_n=kw["c_n"])
            elif kind == "pass" and self.cfg:
                ops.sched_step(2, ws["x0"], None, ws["x0"], n, x0u=x0u, cfg_scale=kw.get("cfg_scale", 0.0))
        prog.hold(*[v for v in kw.values() if torch.is_tensor(v)])
        self._progs[key] = prog
        return prog

    def _set_steps(self, ts):
        if len(ts) > self.tseq.numel():
            self.tseq = torch.zeros(len(ts), dtype=torch.int32, device=self.device)
            self._progs = {}
        self.tseq[: len(ts)].copy_(torch.tensor(ts, dtype=torch.int32))
        self.step.zero_()

    def _run(self, prog, n_steps, use_graph):
        if use_graph:
            prog.instantiate()
            prog.replay(n_steps)
        else:
            for _ in range(n_steps):
                prog.run()

    # ------------------------------------------------------------------------------------------
    def denoise(self, x, t, cfg_scale=2.5):
        """One FDM.forward: x [B, L*G, c] -> x0_hat [B, L*G, c] (CFG-mixed when prepared with cfg=True)."""
        self._check_x(x)
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.ws["x"].copy_(x.reshape(self.M, self.p.d))
            self._set_steps([int(t)])
            self._program("pass", cfg_scale=float(cfg_scale)).run()
            out = self.ws["x0"][: self.M].clone().reshape(x.shape)
        cur.wait_stream(self.stream)
        return out

    def _check_x(self, x):
        if self.M == 0:
            raise FdmError("call prepare() first")
        if tuple(x.shape) != (self.B, self.L * self.p.G, self.p.c):
            raise FdmError(f"latent shape {tuple(x.shape)} != {(self.B, self.L * self.p.G, self.p.c)}")

    def sample_ddpm(self, x_T, t_list, noise=None, seed=0, clip0=0, cfg_scale=2.5, use_graph=True, record=None):
        """p_sample_loop over t_list (descending).  noise [len(t_list), B, L*G, c] injects z per step;
        otherwise z is drawn in-kernel (Philox keyed by seed, global clip index clip0 + b, step)."""
        self._check_x(x_T)
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.ws["x"].copy_(x_T.reshape(self.M, self.p.d))
            self._set_steps(list(t_list))
            kw = dict(cfg_scale=float(cfg_scale), clip0=int(clip0))
            if noise is not None:
                kw["noise"] = _dev(noise, self.device)
            else:
                kw["seed"] = int(seed)
            prog = self._program("ddpm", **kw)
            if record is None:
                self._run(prog, len(t_list), use_graph)
            else:
                for _ in t_list:
                    self._run(prog, 1, use_graph)
                    record.append(self.ws["x"].clone().reshape(x_T.shape))
            out = self.ws["x"].clone().reshape(x_T.shape)
        cur.wait_stream(self.stream)
        return out

    def sample_ddim(self, x_T, steps, cfg_scale=2.5, use_graph=True):
        """ddim_sample (eta = 0).  The last pair (t, -1) never updates the latent in the reference
        (:695-696), so its denoiser call is skipped: exact."""
        self._check_x(x_T)
        if steps not in self._ddim:
            pairs = [pr for pr in schedule.ddim_time_pairs(steps) if pr[1] >= 0]
            san, cn = schedule.ddim_tables(schedule.make_buffers(1000), pairs)
            self._ddim[steps] = (pairs, san.to(self.device), cn.to(self.device))
        pairs, san, cn = self._ddim[steps]
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.ws["x"].copy_(x_T.reshape(self.M, self.p.d))
            self._set_steps([pr[0] for pr in pairs])
            prog = self._program("ddim", cfg_scale=float(cfg_scale), sqrt_an=san, c_n=cn)
            self._run(prog, len(pairs), use_graph)
            out = self.ws["x"].clone().reshape(x_T.shape)
        cur.wait_stream(self.stream)
        return out
