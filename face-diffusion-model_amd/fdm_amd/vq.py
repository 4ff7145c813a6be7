"""(E)VQ-VAE quantise + decode on the HIP path.

quant : models/lib/quantizer.py:35-64, models/vq_vae_emotion.py:221-252 -> fdm_op_vq_quant
decode: models/vq_vae_vocaset.py:35-43,245-258 / models/vq_vae_emotion.py:33-41,335-352 /
        models/vq_vae.py:275-347 with models/lib/base_models.py:37-87,138-174,286-301:
  [decoder_linear_embedding_pre GEMM]                       (MEAD / BIWI only)
  Conv1d(1024, 1024, k=5, replicate pad)   fdm_op_pad_rows + fdm_op_gemm with overlapping rows (K = 5*1024)
  LeakyReLU(0.2) + InstanceNorm1d          fdm_op_leaky_instnorm
  Linear + pe[0] (the reference indexes the positional table by batch position, base_models.py:300;
                  bs = 1 usage == pe[0] for every clip)   fdm_op_gemm, row-broadcast residual
  6 x pre-LN {attention (scale = hidden^-0.5, base_models.py:144), tanh-GELU MLP}
  Linear(1024, V3)                         fdm_op_gemm (N = 15069 / 70110, unaligned rows handled)
"""
import math

import torch

from . import ops, presets
from ._lib import ACT_GELU_TANH, ACT_LEAKY02, ACT_NONE, BF16, F32, FdmError
from .presets import VQ_FFN, VQ_HEADS, VQ_HIDDEN, VQ_LAYERS


class VQPlan:
    @torch.inference_mode(False)      # plan state must stay writable outside a caller's inference_mode block
    def __init__(self, preset, weights, dtype=F32, device="cuda:0"):
        self.p = presets.get(preset)
        self.dtype, self.td = dtype, ops.tdtype(dtype)
        self.device = dv = torch.device(device)
        self.stream = torch.cuda.Stream(device=dv)
        g = lambda k: weights[k].detach().to(device=dv, dtype=torch.float32).contiguous()
        p, d = self.p, VQ_HIDDEN
        with torch.cuda.stream(self.stream):
            op = lambda t: ops.to_operand(t.contiguous(), dtype)
            self.codebook = g("quantize.embedding.weight")
            if self.codebook.shape != (p.K * p.n_books, p.c):
                raise FdmError(f"codebook shape {tuple(self.codebook.shape)} != {(p.K * p.n_books, p.c)}")
            self.pre = None
            if p.vq_pre:
                self.pre = (op(g("decoder.decoder_linear_embedding_pre.net.weight")), g("decoder.decoder_linear_embedding_pre.net.bias"))
            wc = g("decoder.expander.0.0.weight")                                   # [1024, 1024, 5]
            self.conv_w = op(wc.permute(0, 2, 1).reshape(d, 5 * d))                  # [out, (k, in)]
            self.conv_b = g("decoder.expander.0.0.bias")
            self.emb = (op(g("decoder.decoder_linear_embedding.net.weight")), g("decoder.decoder_linear_embedding.net.bias"))
            pe0 = torch.zeros(d)
            pe0[1::2] = 1.0                                                          # sin(0) = 0, cos(0) = 1
            self.pe0 = pe0.to(dv).view(1, d)
            def blocks(prefix):
                out = []
                for l in range(VQ_LAYERS):
                    a = f"{prefix}.net.{2 * l}.fn."
                    m = f"{prefix}.net.{2 * l + 1}.fn."
                    out.append(dict(
                        ln1=(g(a + "norm.weight"), g(a + "norm.bias")), wqkv=op(g(a + "fn.to_qkv.weight")),
                        wo=op(g(a + "fn.to_out.weight")), bo=g(a + "fn.to_out.bias"),
                        ln2=(g(m + "norm.weight"), g(m + "norm.bias")),
                        w1=op(g(m + "fn.l1.weight")), b1=g(m + "fn.l1.bias"), w2=op(g(m + "fn.l2.weight")), b2=g(m + "fn.l2.bias")))
                return out
            self.layers = blocks("decoder.decoder_transformer")
            self.out_w = op(g("decoder.vertice_map_reverse.weight"))
            self.out_b = g("decoder.vertice_map_reverse.bias") if "decoder.vertice_map_reverse.bias" in weights else None
            # encoder (models/vq_vae_vocaset.py:134-191): optional, only needed for encode()
            self.enc = None
            if "encoder.vertice_mapping.0.weight" in weights:
                wm = g("encoder.vertice_mapping.0.weight")
                self.Kp = (p.V3 + 63) // 64 * 64           # K of the GEMM must be a multiple of the k-tile: zero-pad once
                wmp = torch.zeros(d, self.Kp, device=dv)
                wmp[:, : p.V3] = wm
                we = g("encoder.squasher.0.0.weight")
                self.enc = dict(
                    map_w=op(wmp), map_b=g("encoder.vertice_mapping.0.bias"),
                    emo=(g("encoder.emotion_mapping.0.weight"), g("encoder.emotion_mapping.0.bias")) if p.n_books > 1 else None,
                    conv_w=op(we.permute(0, 2, 1).reshape(d, 5 * d)), conv_b=g("encoder.squasher.0.0.bias"),
                    emb=(op(g("encoder.encoder_linear_embedding.net.weight")), g("encoder.encoder_linear_embedding.net.bias")),
                    post=(op(g("encoder.encoder_linear_embedding_post.net.weight")), g("encoder.encoder_linear_embedding_post.net.bias"))
                    if p.vq_pre else None,
                    layers=blocks("encoder.encoder_transformer"))
        self.stream.synchronize()

    def _transformer(self, h, layers, B, L):
        """6 pre-LN blocks on the fp32 residual stream h [B*L, 1024] (updated in place and returned)."""
        dv, td, dt, d = self.device, self.td, self.dtype, VQ_HIDDEN
        M, H, hd = B * L, VQ_HEADS, VQ_HIDDEN // VQ_HEADS
        z = lambda *s, dtp=torch.float32: torch.empty(*s, device=dv, dtype=dtp)
        q = z(M, d, dtp=td)
        kp, vp, Lpad = ops.kv_buffers(B, H, L, hd, td, dv)
        ctx, u, hb, a = z(M, d, dtp=td), z(M, VQ_FFN, dtp=td), z(M, d), z(M, d, dtp=td)
        for ly in layers:
            ops.layernorm(h, ly["ln1"][0], ly["ln1"][1], M, d, y_t=a, dtype=dt)
            ops.gemm(a, ly["wqkv"], M, 3 * d, d, out_t=q, ldo_t=d, out_kp=kp, kp_col0=d, out_vp=vp, vp_col0=2 * d,
                     kv_L=L, kv_Lpad=Lpad, kv_hd=hd)
            ops.attention(q, kp, vp, ctx, B=B, H=H, L=L, hd=hd, ldq=d, ldo=d, Lpad=Lpad,
                          scale=d ** -0.5, causal=False)
            ops.gemm(ctx, ly["wo"], M, d, d, bias=ly["bo"], resid=h, out_f32=hb)
            ops.layernorm(hb, ly["ln2"][0], ly["ln2"][1], M, d, y_t=a, dtype=dt)
            ops.gemm(a, ly["w1"], M, VQ_FFN, d, bias=ly["b1"], act=ACT_GELU_TANH, out_t=u)
            ops.gemm(u, ly["w2"], M, d, VQ_FFN, bias=ly["b2"], resid=hb, out_f32=h)
        return h

    def _conv_norm_embed(self, xt, conv_w, conv_b, emb, B, L):
        """Conv1d(k=5, replicate) -> LeakyReLU -> InstanceNorm1d -> Linear + pe[0]; xt [B*L, 1024] operand dtype."""
        dv, td, dt, d = self.device, self.td, self.dtype, VQ_HIDDEN
        M = B * L
        z = lambda *s, dtp=torch.float32: torch.empty(*s, device=dv, dtype=dtp)
        xp = z(B, L + 4, d, dtp=td)
        ops.pad_rows(xt, xp, B, L, d, 2)
        c32 = z(M, d)
        ops.gemm(xp, conv_w, L, d, 5 * d, lda=d, bias=conv_b, out_f32=c32, batch=B, a_bs=(L + 4) * d, out_bs=L * d)
        nt = z(M, d, dtp=td)
        ops.leaky_instnorm(c32, B, L, d, y_t=nt, dtype=dt)
        h = z(M, d)
        ops.gemm(nt, emb[0], M, d, d, bias=emb[1], resid=self.pe0, ldr=d, resid_row_mod=1, out_f32=h)
        return h

    def encode(self, x, emo=None):
        """x [B, L, V3] fp32 (vertices minus template) -> latent [B, L*G, c] fp32 (VQAutoEncoder.encode)."""
        if self.enc is None:
            raise FdmError("this plan was built without encoder.* weights")
        p, dv, td, dt, d = self.p, self.device, self.td, self.dtype, VQ_HIDDEN
        B, L, V3 = x.shape
        if V3 != p.V3 or L < 2:
            raise FdmError(f"bad vertex tensor shape {tuple(x.shape)}")
        M, e = B * L, self.enc
        cur = torch.cuda.current_stream(dv)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            x32 = torch.zeros(M, self.Kp, device=dv)                              # zero-padded K (layout only)
            x32[:, :V3] = x.detach().to(device=dv, dtype=torch.float32).reshape(M, V3)
            xp = ops.to_operand(x32, dt)
            h = torch.empty(M, d, device=dv)
            ops.gemm(xp, e["map_w"], M, d, self.Kp, bias=e["map_b"], act=ACT_LEAKY02, out_f32=h)
            if e["emo"] is not None:
                if emo is None:
                    raise FdmError("this preset's encoder needs the emotion one-hot")
                emo = emo.to(dv).float()
                emo = (emo.unsqueeze(0).expand(B, -1) if emo.dim() == 1 else emo.reshape(B, -1)).contiguous()
                em = torch.empty(B, d, device=dv)
                ops.small_linear(emo, e["emo"][0], e["emo"][1], em, B, 7, d, ACT_LEAKY02)
                h2 = torch.empty(M, d, device=dv)
                ops.add_rows(h2, M, d, h, 1, M, em, L, B)
                h = h2
            h = self._conv_norm_embed(ops.to_operand(h, dt), e["conv_w"], e["conv_b"], e["emb"], B, L)
            h = self._transformer(h, e["layers"], B, L)
            if e["post"] is not None:
                out = torch.empty(M, p.G * p.c, device=dv)
                ops.gemm(ops.to_operand(h, dt), e["post"][0], M, p.G * p.c, d, bias=e["post"][1], out_f32=out)
            else:
                out = h
        cur.wait_stream(self.stream)
        return out.view(B, L * p.G, p.c)

    # ------------------------------------------------------------------------------------------
    def quant(self, z, emo=None):
        """z [B, R, c] fp32 -> (z_q [B, c, R] fp32, idx [B*R, 1] int64), book chosen by argmax(one_hot)."""
        p, dv = self.p, self.device
        z = z.detach().to(device=dv, dtype=torch.float32).contiguous()
        B, R, c = z.shape
        if c != p.c:
            raise FdmError(f"latent width {c} != zquant_dim {p.c}")
        book = None
        if p.n_books > 1:
            if emo is None:
                raise FdmError("this preset needs the emotion one-hot to pick the codebook slice")
            emo = emo.to(dv)
            if emo.dim() == 1:
                emo = emo.unsqueeze(0).expand(B, -1)
            book = torch.argmax(emo, dim=1).to(torch.int32).contiguous()
        zq = torch.empty(B, c, R, device=dv)
        idx = torch.empty(B * R, 1, device=dv, dtype=torch.int64)
        ops.vq_quant(z, self.codebook, book, B, R, c, p.K, zq, idx)
        return zq, idx

    def decode(self, zq):
        """zq [B, c, L*G] -> vertex offsets [B, L, V3] fp32 (template is added by the caller)."""
        p, dv, td, dt, d = self.p, self.device, self.td, self.dtype, VQ_HIDDEN
        B, c, R = zq.shape
        if c != p.c or R % p.G:
            raise FdmError(f"bad quantised latent shape {tuple(zq.shape)}")
        L = R // p.G
        if L < 2:
            raise FdmError("decode needs at least 2 frames (InstanceNorm1d over one element is undefined in the reference)")
        M = B * L
        H, hd = VQ_HEADS, d // VQ_HEADS
        z = lambda *s, dtp=torch.float32: torch.empty(*s, device=dv, dtype=dtp)
        cur = torch.cuda.current_stream(dv)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            x = zq.detach().to(device=dv, dtype=torch.float32).permute(0, 2, 1).reshape(M, p.G * p.c).contiguous()   # layout only
            xt = ops.to_operand(x, dt)
            if self.pre is not None:
                y = z(M, d, dtp=td)
                ops.gemm(xt, self.pre[0], M, d, p.G * p.c, bias=self.pre[1], out_t=y)
                xt = y
            elif p.G * p.c != d:
                raise FdmError("decoder input width must equal hidden size when there is no pre-embedding")
            h = self._conv_norm_embed(xt, self.conv_w, self.conv_b, self.emb, B, L)
            h = self._transformer(h, self.layers, B, L)
            ht = ops.to_operand(h, dt)
            out = z(M, p.V3)
            ops.gemm(ht, self.out_w, M, p.V3, d, bias=self.out_b, out_f32=out)
        cur.wait_stream(self.stream)
        return out.view(B, L, p.V3)
