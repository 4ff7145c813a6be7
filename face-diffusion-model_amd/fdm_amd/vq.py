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
import torch

from . import presets
from ._lib import F32, FdmError


class VQPlan:
    """Thin binding of the library's VQ object (include/fdm_hip.h: fdm_vq_create / _set_weights / _quant / _decode / _encode;
    implementation csrc/encoders.hip)."""

    def __init__(self, preset, weights, dtype=F32, device="cuda:0"):
        import ctypes as C
        from ._lib import VqDesc, check, lib
        self.p = p = presets.get(preset)
        self.dtype = dtype
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise FdmError("VQPlan runs on the HIP path only (no CPU fallback)")
        cb = weights["quantize.embedding.weight"]
        if tuple(cb.shape) != (p.K * p.n_books, p.c):
            raise FdmError(f"codebook shape {tuple(cb.shape)} != {(p.K * p.n_books, p.c)}")
        self.has_encoder = "encoder.vertice_mapping.0.weight" in weights
        self.h = None
        h = C.c_void_p()
        desc = VqDesc(p.G, p.c, p.K, p.n_books, p.V3, int(p.vq_pre))
        check(lib().fdm_vq_create(C.byref(desc), dtype, C.byref(h)))
        self.h = h
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream().cuda_stream
            keep = []
            for k, v in weights.items():
                if not (k.startswith("quantize.") or k.startswith("decoder.") or k.startswith("encoder.")) or k.endswith(".pe"):
                    continue
                t = v.detach().to(torch.float32).contiguous()
                keep.append(t)
                check(lib().fdm_vq_set_weights(h, k.encode(), t.data_ptr(), t.numel(), st))
            torch.cuda.current_stream().synchronize()

    def __del__(self):
        try:
            if self.h:
                from ._lib import lib
                lib().fdm_vq_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _emo(self, emo, B, n):
        if emo is None:
            return None
        emo = emo.detach().to(device=self.device, dtype=torch.float32)
        return (emo.unsqueeze(0).expand(B, -1) if emo.dim() == 1 else emo.reshape(B, -1)).contiguous()

    def encode(self, x, emo=None):
        """x [B, L, V3] fp32 (vertices minus template) -> latent [B, L*G, c] fp32 (VQAutoEncoder.encode)."""
        from ._lib import check, lib
        if not self.has_encoder:
            raise FdmError("this plan was built without encoder.* weights")
        p, dv = self.p, self.device
        B, L, V3 = x.shape
        if V3 != p.V3 or L < 2:
            raise FdmError(f"bad vertex tensor shape {tuple(x.shape)}")
        if p.n_books > 1 and emo is None:
            raise FdmError("this preset's encoder needs the emotion one-hot")
        x = x.detach().to(device=dv, dtype=torch.float32).contiguous()
        em = self._emo(emo, B, 7) if p.n_books > 1 else None
        out = torch.empty(B, L * p.G, p.c, device=dv)
        with torch.cuda.device(dv):
            check(lib().fdm_vq_encode(self.h, x.data_ptr(), em.data_ptr() if em is not None else None, B, L, out.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream))
        self._in = (x, em)
        return out

    def quant(self, z, emo=None):
        """z [B, R, c] fp32 -> (z_q [B, c, R] fp32, idx [B*R, 1] int64), book chosen by argmax(one_hot)."""
        from ._lib import check, lib
        p, dv = self.p, self.device
        z = z.detach().to(device=dv, dtype=torch.float32).contiguous()
        B, R, c = z.shape
        if c != p.c:
            raise FdmError(f"latent width {c} != zquant_dim {p.c}")
        if p.n_books > 1 and emo is None:
            raise FdmError("this preset needs the emotion one-hot to pick the codebook slice")
        em = self._emo(emo, B, p.n_books) if p.n_books > 1 else None
        zq = torch.empty(B, c, R, device=dv)
        idx = torch.empty(B * R, 1, device=dv, dtype=torch.int64)
        with torch.cuda.device(dv):
            check(lib().fdm_vq_quant(self.h, z.data_ptr(), em.data_ptr() if em is not None else None, B, R, zq.data_ptr(), idx.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream))
        self._in = (z, em)
        return zq, idx

    def quant_full(self, z, emo=None, beta=0.25, min_encodings=True):
        """The reference's whole quant() tuple (models/lib/quantizer.py:35-64, models/vq_vae_vocaset.py:16-18,31-33):
        (z_q [B, c, R], emb_loss, (perplexity, min_encodings [B*R, 256], indices [B*R, 1]))."""
        from ._lib import check, lib
        zq, idx = self.quant(z, emo)
        z_, em = self._in
        B, R, _ = z_.shape
        dv = self.device
        me = torch.empty(B * R, self.p.K, device=dv) if min_encodings else None
        out2 = torch.empty(2, device=dv)
        with torch.cuda.device(dv):
            check(lib().fdm_vq_quant_stats(self.h, z_.data_ptr(), em.data_ptr() if em is not None else None, idx.data_ptr(), B, R,
                                           float(beta), me.data_ptr() if me is not None else None, out2.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream))
        return zq, out2[0], (out2[1], me, idx)

    def decode(self, zq):
        """zq [B, c, L*G] -> vertex offsets [B, L, V3] fp32 (template is added by the caller)."""
        from ._lib import check, lib
        p, dv = self.p, self.device
        B, c, R = zq.shape
        if c != p.c or R % p.G:
            raise FdmError(f"bad quantised latent shape {tuple(zq.shape)}")
        L = R // p.G
        if L < 2:
            raise FdmError("decode needs at least 2 frames (InstanceNorm1d over one element is undefined in the reference)")
        zq = zq.detach().to(device=dv, dtype=torch.float32).contiguous()
        out = torch.empty(B, L, p.V3, device=dv)
        with torch.cuda.device(dv):
            check(lib().fdm_vq_decode(self.h, zq.data_ptr(), B, R, out.data_ptr(), torch.cuda.current_stream().cuda_stream))
        self._in = (zq,)
        return out
