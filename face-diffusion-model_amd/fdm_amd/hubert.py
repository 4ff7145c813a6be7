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

from ._lib import F32, FdmError

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


class HubertPlan:
    """Thin binding of the library's audio-encoder object (include/fdm_hip.h: fdm_hubert_create / _set_weights / _forward;
    implementation csrc/encoders.hip).  Weights go in by transformers state-dict name and are repacked on the device;
    forward() only passes pointers."""

    def __init__(self, weights, n_layers=None, dtype=F32, device="cuda:0", prefix="", cfg=HUBERT_LARGE):
        import ctypes as C
        from ._lib import check, lib
        self.cfg = cfg
        self.dtype = dtype
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise FdmError("HubertPlan runs on the HIP path only (no CPU fallback)")
        self.n_layers = cfg.n_layers if n_layers is None else n_layers
        self.h = None
        h = C.c_void_p()
        kind = 0 if cfg.stable_ln else 1
        check(lib().fdm_hubert_create(kind, self.n_layers, dtype, C.byref(h)))
        self.h = h
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream().cuda_stream
            keep = []
            for k, v in weights.items():
                if prefix and not k.startswith(prefix):
                    continue
                k = k[len(prefix):]
                if not (k.startswith("feature_extractor.") or k.startswith("feature_projection.") or k.startswith("encoder.")):
                    continue
                if k.startswith("encoder.layers."):
                    if int(k.split(".")[2]) >= self.n_layers:
                        continue
                t = v.detach().to(torch.float32).contiguous()
                keep.append(t)
                check(lib().fdm_hubert_set_weights(h, k.encode(), t.data_ptr(), t.numel(), st))
            torch.cuda.current_stream().synchronize()

    def __del__(self):
        try:
            if self.h:
                from ._lib import lib
                lib().fdm_hubert_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def forward(self, wav, frame_num=None, interp_fps=None):
        """wav [B, n] fp32 (processor-normalised) -> last_hidden_state [B, N, D] fp32.
        frame_num: keep at most 2*frame_num conv frames (models/hubert.py:97-98).  interp_fps=(in, out): resample the
        conv features (linear, align_corners) to frame_num / int(T/in*out) frames instead of the even crop."""
        import ctypes as C
        from ._lib import check, lib
        dv = self.device
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
        if interp_fps:
            N = int(frame_num) if frame_num else int(Ts[-1] / float(interp_fps[0]) * interp_fps[1])
            if N < 2:
                raise FdmError(f"interpolated length {N} too short")
        out = torch.empty(B, N, self.cfg.D, device=dv)
        nf = C.c_int(0)
        with torch.cuda.device(dv):
            check(lib().fdm_hubert_forward(self.h, wav.data_ptr(), B, n, int(frame_num or 0), int(interp_fps[0]) if interp_fps else 0,
                                           int(interp_fps[1]) if interp_fps else 0, out.data_ptr(), C.byref(nf),
                                           torch.cuda.current_stream().cuda_stream))
        if nf.value != N:
            raise FdmError(f"fdm_hubert_forward produced {nf.value} frames, expected {N}")
        self._wav = wav          # read asynchronously on this stream
        return out
