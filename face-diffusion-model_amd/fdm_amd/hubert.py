"""HuBERT-large audio encoder on the HIP path (models/hubert.py:75-146 over transformers'
HubertFeatureEncoder / HubertFeatureProjection / HubertEncoderStableLayerNorm).

Run once per clip: it does not depend on (t, x_t), so the reference's per-step re-run
(models/fdm_vocaset.py:59) is hoisted (exact).

Kernel mapping (all through the C ABI):
  conv layer 0 (C_in = 1, k = 10, s = 5)      fdm_op_conv0, channels-last output [B, T0, 512]
  LayerNorm(512) + GELU(erf)                   fdm_op_layernorm (act fused)
  conv layers 1..6 (k = 3,3,3,3,2,2; s = 2)    fdm_op_gemm with overlapping A rows: row t of the im2col
                                               matrix is the contiguous slice x[t*s : t*s + k, :] of the
                                               channels-last signal (lda = s*512, K = k*512), weights
                                               repacked once to [out, k, in]
  drop last frame if odd (:95-96)              per-clip batch stride, M = N
  feature projection LN + Linear(512, 1024)    fdm_op_layernorm, fdm_op_gemm
  positional conv (k = 128, groups = 16)       fdm_op_group_pad -> 16 grouped GEMMs (lda = 64, K = 8192)
                                               with bias + GELU + residual fused
  24 x pre-LN layer                            LN -> QKV GEMM (V^T scatter) -> fused attention (hd 64,
                                               non-causal) -> out-proj GEMM + residual -> LN -> FFN1 GEMM +
                                               GELU -> FFN2 GEMM + residual
  final LayerNorm                              fdm_op_layernorm
"""
from dataclasses import dataclass

import torch

from . import ops
from ._lib import ACT_GELU_ERF, ACT_NONE, BF16, F32, FdmError

CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)
CD = 512
POS_K, POS_G = 128, 16


@dataclass(frozen=True)
class AudioEncoderConfig:
    """The two audio encoders of the reference share one op graph up to these switches."""
    name: str
    D: int            # hidden size
    H: int            # attention heads (head_dim 64 for both)
    FFN: int
    n_layers: int
    conv_norm: str    # 'layer': LayerNorm(512) after every conv (HuBERT-large) | 'group': GroupNorm on conv 0 only
    conv_bias: bool
    stable_ln: bool   # True: pre-LN layers + final LayerNorm | False: LayerNorm before the stack, post-LN layers


HUBERT_LARGE = AudioEncoderConfig("hubert-large", 1024, 16, 4096, 24, "layer", True, True)
# models/wav2vec.py:69-143 over transformers Wav2Vec2Model ('wav2vec2-base-960h' config): the BIWI audio encoder
WAV2VEC2_BASE = AudioEncoderConfig("wav2vec2-base", 768, 12, 3072, 12, "group", False, False)
N_HEAD, HD, D, FFN = 16, 64, 1024, 4096          # HuBERT-large constants (kept for callers)


def conv_lengths(n):
    out = []
    for k, s in zip(CONV_KERNEL, CONV_STRIDE):
        n = (n - k) // s + 1
        out.append(n)
    return out


def num_frames(n_samples):
    t = conv_lengths(n_samples)[-1]
    return t - (t % 2)


def _get(w, name):
    """weight_g / weight_v (torch 2.0) and parametrizations.weight.original0/1 (torch >= 2.1) are both accepted."""
    if name in w:
        return w[name]
    alt = {"encoder.pos_conv_embed.conv.parametrizations.weight.original0": "encoder.pos_conv_embed.conv.weight_g",
           "encoder.pos_conv_embed.conv.parametrizations.weight.original1": "encoder.pos_conv_embed.conv.weight_v"}
    if name in alt and alt[name] in w:
        return w[alt[name]]
    raise FdmError(f"missing HuBERT weight {name}")


class HubertPlan:
    @torch.inference_mode(False)      # plan state must stay writable outside a caller's inference_mode block
    def __init__(self, weights, n_layers=None, dtype=F32, device="cuda:0", prefix="", cfg=HUBERT_LARGE):
        self.cfg = cfg
        n_layers = cfg.n_layers if n_layers is None else n_layers
        D = cfg.D
        self.dtype, self.td = dtype, ops.tdtype(dtype)
        self.device = dv = torch.device(device)
        self.n_layers = n_layers
        g = lambda k: _get(weights, prefix + k).detach().to(device=dv, dtype=torch.float32).contiguous()
        self.stream = torch.cuda.Stream(device=dv)
        with torch.cuda.stream(self.stream):
            op = lambda t: ops.to_operand(t.contiguous(), dtype)
            self.conv = []
            for i, k in enumerate(CONV_KERNEL):
                p = f"feature_extractor.conv_layers.{i}."
                wt = g(p + "conv.weight")
                if i == 0:
                    wk = wt.reshape(CD, k).contiguous()                       # fp32, direct kernel
                else:
                    wk = op(wt.permute(0, 2, 1).reshape(CD, k * CD))          # [out, (k, in)]
                has_norm = cfg.conv_norm == "layer" or i == 0
                self.conv.append((wk, g(p + "conv.bias") if cfg.conv_bias else None,
                                  g(p + "layer_norm.weight") if has_norm else None,
                                  g(p + "layer_norm.bias") if has_norm else None))
            self.fp_ln = (g("feature_projection.layer_norm.weight"), g("feature_projection.layer_norm.bias"))
            self.fp_w, self.fp_b = op(g("feature_projection.projection.weight")), g("feature_projection.projection.bias")
            # weight-normalised grouped positional conv (weight_norm dim = 2), repacked per group
            wg = g("encoder.pos_conv_embed.conv.parametrizations.weight.original0")
            wv = g("encoder.pos_conv_embed.conv.parametrizations.weight.original1")
            wpc = wg * wv / wv.norm(2, dim=(0, 1), keepdim=True)                   # [1024, 64, 128]
            dg = D // POS_G
            self.pc_w = op(wpc.view(POS_G, dg, dg, POS_K).permute(0, 1, 3, 2).reshape(POS_G, dg, POS_K * dg))
            self.pc_b = g("encoder.pos_conv_embed.conv.bias")
            self.layers = []
            for l in range(n_layers):
                p = f"encoder.layers.{l}."
                wqkv = torch.cat([g(p + "attention.q_proj.weight"), g(p + "attention.k_proj.weight"), g(p + "attention.v_proj.weight")])
                bqkv = torch.cat([g(p + "attention.q_proj.bias"), g(p + "attention.k_proj.bias"), g(p + "attention.v_proj.bias")])
                self.layers.append(dict(
                    ln1=(g(p + "layer_norm.weight"), g(p + "layer_norm.bias")), wqkv=op(wqkv), bqkv=bqkv.contiguous(),
                    wo=op(g(p + "attention.out_proj.weight")), bo=g(p + "attention.out_proj.bias"),
                    ln2=(g(p + "final_layer_norm.weight"), g(p + "final_layer_norm.bias")),
                    w1=op(g(p + "feed_forward.intermediate_dense.weight")), b1=g(p + "feed_forward.intermediate_dense.bias"),
                    w2=op(g(p + "feed_forward.output_dense.weight")), b2=g(p + "feed_forward.output_dense.bias")))
            self.final_ln = (g("encoder.layer_norm.weight"), g("encoder.layer_norm.bias"))
        self.stream.synchronize()

    def forward(self, wav, frame_num=None, interp_fps=None):
        """wav [B, n] fp32 (processor-normalised) -> last_hidden_state [B, N, D] fp32.
        frame_num: keep at most 2*frame_num conv frames (models/hubert.py:97-98).  interp_fps=(in, out): resample the
        conv features (linear, align_corners) to frame_num / int(T/in*out) frames instead of the even crop."""
        dv, td, dt, cfg = self.device, self.td, self.dtype, self.cfg
        D, N_HEAD, FFN = cfg.D, cfg.H, cfg.FFN
        if wav.dim() == 1:
            wav = wav.unsqueeze(0)
        wav = wav.detach().to(device=dv, dtype=torch.float32).contiguous()
        B, n = wav.shape
        Ts = conv_lengths(n)
        if Ts[-1] < 2:
            raise FdmError(f"audio too short: {n} samples")
        N = Ts[-1] - (Ts[-1] % 2)
        if frame_num and not interp_fps and N > frame_num * 2:
            N = frame_num * 2
        cur = torch.cuda.current_stream(dv)
        self.stream.wait_stream(cur)
        z = lambda *s, dtp=torch.float32: torch.empty(*s, device=dv, dtype=dtp)
        with torch.cuda.stream(self.stream):
            # --- conv feature extractor ---
            x32 = z(B * Ts[0], CD)
            ops.conv0(wav, self.conv[0][0], self.conv[0][1], x32, B, n, Ts[0])
            xt = z(B * Ts[0], CD, dtp=td)
            if cfg.conv_norm == "layer":
                ops.layernorm(x32, self.conv[0][2], self.conv[0][3], B * Ts[0], CD, act=ACT_GELU_ERF, y_t=xt, dtype=dt)
            else:   # GroupNorm(512 groups) over time, affine, then GELU (Wav2Vec2GroupNormConvLayer)
                ops.time_groupnorm(x32, self.conv[0][2], self.conv[0][3], B, Ts[0], CD, y_t=xt, act=ACT_GELU_ERF, dtype=dt)
            Tin = Ts[0]
            g6 = None
            for i in range(1, 7):
                k, s, To = CONV_KERNEL[i], CONV_STRIDE[i], Ts[i]
                if cfg.conv_norm == "layer":
                    y32 = z(B * To, CD)
                    ops.gemm(xt, self.conv[i][0], To, CD, k * CD, lda=s * CD, bias=self.conv[i][1], out_f32=y32,
                             batch=B, a_bs=Tin * CD, out_bs=To * CD)
                    if i < 6:
                        xt = z(B * To, CD, dtp=td)
                        ops.layernorm(y32, self.conv[i][2], self.conv[i][3], B * To, CD, act=ACT_GELU_ERF, y_t=xt, dtype=dt)
                    else:
                        g6 = z(B * To, CD)
                        ops.layernorm(y32, self.conv[i][2], self.conv[i][3], B * To, CD, act=ACT_GELU_ERF, y_f32=g6)
                else:   # conv (no norm) + GELU fused in the GEMM epilogue
                    if i < 6:
                        nx = z(B * To, CD, dtp=td)
                        ops.gemm(xt, self.conv[i][0], To, CD, k * CD, lda=s * CD, bias=self.conv[i][1], act=ACT_GELU_ERF,
                                 out_t=nx, batch=B, a_bs=Tin * CD, out_bs=To * CD)
                        xt = nx
                    else:
                        g6 = z(B * To, CD)
                        ops.gemm(xt, self.conv[i][0], To, CD, k * CD, lda=s * CD, bias=self.conv[i][1], act=ACT_GELU_ERF,
                                 out_f32=g6, batch=B, a_bs=Tin * CD, out_bs=To * CD)
                Tin = To
            # --- even crop (models/hubert.py:95-96) + feature projection ---
            T6 = Ts[6]
            if interp_fps:
                N = int(frame_num) if frame_num else int(T6 / float(interp_fps[0]) * interp_fps[1])
                if N < 2:
                    raise FdmError(f"interpolated length {N} too short")
                gi = z(B * N, CD)
                ops.linear_interp(g6, gi, B, T6, N, CD)
                g6, T6 = gi, N
            ft = z(B * T6, CD, dtp=td)
            ops.layernorm(g6, self.fp_ln[0], self.fp_ln[1], B * T6, CD, y_t=ft, dtype=dt)
            M = B * N
            h = z(M, D)
            ht = z(M, D, dtp=td) if dt == BF16 else h
            ops.gemm(ft, self.fp_w, N, D, CD, bias=self.fp_b, out_f32=h, out_t=ht if dt == BF16 else None,
                     batch=B, a_bs=T6 * CD, out_bs=N * D)
            # --- positional conv embedding: h += GELU(grouped conv(h)) ---
            dg = D // POS_G
            xg = z(POS_G, B, N + POS_K, dg, dtp=td)
            ops.group_pad(ht, xg, B, N, D, POS_G, POS_K // 2)
            h2 = z(M, D)
            for b in range(B):
                ops.gemm(xg[:, b], self.pc_w, N, dg, POS_K * dg, lda=dg, batch=POS_G, a_bs=B * (N + POS_K) * dg,
                         w_bs=dg * POS_K * dg, bias=self.pc_b, bias_bs=dg, act=ACT_GELU_ERF, resid=h[b * N:], ldr=D,
                         out_f32=h2[b * N:], ldo_f32=D, out_bs=dg)
            h = h2
            # --- encoder layers ---
            xt = z(M, D, dtp=td)
            q = z(M, D, dtp=td)
            kp, vp, Lpad = ops.kv_buffers(B, N_HEAD, N, HD, td, dv)
            kv = dict(out_t=q, ldo_t=D, out_kp=kp, kp_col0=D, out_vp=vp, vp_col0=2 * D, kv_L=N, kv_Lpad=Lpad, kv_hd=HD)
            ctx = z(M, D, dtp=td)
            u = z(M, FFN, dtp=td)
            hb = z(M, D)
            if cfg.stable_ln:      # pre-LN layers, final LayerNorm (HubertEncoderStableLayerNorm)
                for ly in self.layers:
                    ops.layernorm(h, ly["ln1"][0], ly["ln1"][1], M, D, y_t=xt, dtype=dt)
                    ops.gemm(xt, ly["wqkv"], M, 3 * D, D, bias=ly["bqkv"], **kv)
                    ops.attention(q, kp, vp, ctx, B=B, H=N_HEAD, L=N, hd=HD, ldq=D, ldo=D, Lpad=Lpad,
                                  scale=HD ** -0.5, causal=False)
                    ops.gemm(ctx, ly["wo"], M, D, D, bias=ly["bo"], resid=h, out_f32=hb)
                    ops.layernorm(hb, ly["ln2"][0], ly["ln2"][1], M, D, y_t=xt, dtype=dt)
                    ops.gemm(xt, ly["w1"], M, FFN, D, bias=ly["b1"], act=ACT_GELU_ERF, out_t=u)
                    ops.gemm(u, ly["w2"], M, D, FFN, bias=ly["b2"], resid=hb, out_f32=h)
                out = z(M, D)
                ops.layernorm(h, self.final_ln[0], self.final_ln[1], M, D, y_f32=out)
            else:                  # LayerNorm before the stack, post-LN layers (Wav2Vec2Encoder / Wav2Vec2EncoderLayer)
                x1 = z(M, D)
                both = dt == BF16
                ht = xt if both else None
                ops.layernorm(h, self.final_ln[0], self.final_ln[1], M, D, y_f32=hb, y_t=ht, dtype=dt)
                for ly in self.layers:
                    a_in = xt if both else hb
                    ops.gemm(a_in, ly["wqkv"], M, 3 * D, D, bias=ly["bqkv"], **kv)
                    ops.attention(q, kp, vp, ctx, B=B, H=N_HEAD, L=N, hd=HD, ldq=D, ldo=D, Lpad=Lpad,
                                  scale=HD ** -0.5, causal=False)
                    ops.gemm(ctx, ly["wo"], M, D, D, bias=ly["bo"], resid=hb, out_f32=x1)
                    ops.layernorm(x1, ly["ln1"][0], ly["ln1"][1], M, D, y_f32=hb, y_t=ht, dtype=dt)
                    ops.gemm(a_in, ly["w1"], M, FFN, D, bias=ly["b1"], act=ACT_GELU_ERF, out_t=u)
                    ops.gemm(u, ly["w2"], M, D, FFN, bias=ly["b2"], resid=hb, out_f32=x1)
                    ops.layernorm(x1, ly["ln2"][0], ly["ln2"][1], M, D, y_f32=hb, y_t=ht, dtype=dt)
                out = hb
        cur.wait_stream(self.stream)
        return out.view(B, N, D)
