"""Thin tensor-level wrappers over the C ABI (raw pointers + the current HIP stream).

torch is used only as the owner of device memory and of the stream; every arithmetic operation on
the product path is a libfdm_hip.so kernel.  All wrappers raise if a tensor is not a contiguous
CUDA(HIP) tensor: there is no CPU fallback."""
import ctypes as C

import torch

from . import _lib
from ._lib import (ACT_GELU_ERF, ACT_GELU_TANH, ACT_LEAKY02, ACT_MISH, ACT_NONE, ACT_RELU, BF16, F16, F16X3, F32,
                   AttnArgs, GemmArgs, LnArgs, SchedArgs, check, lib)


class Split:
    """A split-precision GEMM operand (include/fdm_hip.h, FDM_F16X3): planes[0] = hi, planes[1] = lo of a
    [rows, cols] matrix held in one [2, rows, cols] 16-bit tensor.  Slicing rows ([r0:]) keeps the plane distance."""

    def __init__(self, planes, code, row0=0, col0=0):
        self.planes, self.code, self.row0, self.col0 = planes, code, row0, col0
        self.lo_off = planes[0].numel()

    @classmethod
    def empty(cls, rows, cols, code, device):
        return cls(torch.zeros(2, rows, cols, device=device, dtype=tdtype(code)), code)

    def __getitem__(self, sl):
        if not isinstance(sl, slice) or sl.stop is not None or sl.step is not None:
            raise _lib.FdmError("Split operands support [r0:] row offsets only")
        return Split(self.planes, self.code, self.row0 + (sl.start or 0), self.col0)

    @property
    def is_cuda(self):
        return self.planes.is_cuda

    @property
    def shape(self):
        return self.planes.shape[1:]

    def data_ptr(self):
        return self.planes.data_ptr() + (self.row0 * self.planes.shape[2] + self.col0) * self.planes.element_size()

    def numel(self):
        return self.lo_off

    def float(self):
        """hi + lo / SCALE as fp32 (tests)."""
        sc = 2048.0
        return (self.planes[0].float() + self.planes[1].float() / sc)[self.row0:, self.col0:]


def cols(t, c0):
    """The matrix from column c0 on (same row stride: pass lda / ldw = the full width) -- a K range of a GEMM operand."""
    return Split(t.planes, t.code, t.row0, t.col0 + c0) if isinstance(t, Split) else t[:, c0:]


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.FdmError("fdm_amd ops need device tensors (no CPU fallback on the product path)")
    return t.data_ptr()


def _lo(t):
    return t.lo_off if isinstance(t, Split) else 0


def stream():
    return torch.cuda.current_stream().cuda_stream


def tdtype(code):
    """torch dtype of the elements of operand kind `code` (split kinds: of each plane)."""
    return {BF16: torch.bfloat16, F16X3: torch.float16, F16: torch.float16}.get(code, torch.float32)


def is_split(code):
    return code == F16X3


def code_of(t):
    if isinstance(t, Split):
        return t.code
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise _lib.FdmError(f"unsupported dtype {t.dtype}")


def gemm(A, W, M, N, K, *, lda=None, ldw=None, bias=None, act=ACT_NONE, resid=None, ldr=None, resid_row_mod=0,
         out_f32=None, ldo_f32=None, out_t=None, ldo_t=None, batch=1, a_bs=0, w_bs=0, bias_bs=0, out_bs=0,
         out_kp=None, kp_col0=0, out_vp=None, vp_col0=0, kv_L=0, kv_Lpad=0, kv_hd=0,
         stat_out=None, ln_stat_in=None, ln_nparts=0, ln_dim=0, ln_eps=1e-5, ln_colsum=None, rln_gamma=None, rln_beta=None,
         incr_counter=None, incr_table=None, tile=0, sched=None, ksplit=0, ksplit_stride=0,
         batch2=0, a_bs2=0, out_bs2=0):
    a = GemmArgs()
    a.A, a.lda, a.a_batch_stride = _p(A), lda if lda is not None else K, a_bs
    a.W, a.ldw, a.w_batch_stride = _p(W), ldw if ldw is not None else K, w_bs
    a.M, a.N, a.K, a.batch, a.dtype = M, N, K, batch, code_of(A)
    if code_of(W) != a.dtype:
        raise _lib.FdmError("gemm: A and W operand kinds differ")
    a.a_lo_off, a.w_lo_off, a.out_t_lo_off = _lo(A), _lo(W), _lo(out_t)
    a.kv_lo_off = _lo(out_kp) or _lo(out_vp)
    a.bias, a.bias_batch_stride, a.act = _p(bias), bias_bs, act
    a.resid, a.ldr, a.resid_row_mod = _p(resid), (ldr if ldr is not None else N), resid_row_mod
    a.out_f32, a.ldo_f32 = _p(out_f32), (ldo_f32 if ldo_f32 is not None else N)
    a.out_t, a.ldo_t = _p(out_t), (ldo_t if ldo_t is not None else N)
    a.out_batch_stride = out_bs
    a.out_kp, a.kp_col0, a.out_vp, a.vp_col0 = _p(out_kp), kp_col0, _p(out_vp), vp_col0
    a.kv_L, a.kv_Lpad, a.kv_hd = kv_L, kv_Lpad, kv_hd
    a.stat_out, a.ln_stat_in, a.ln_nparts, a.ln_dim, a.ln_eps = _p(stat_out), _p(ln_stat_in), ln_nparts, ln_dim, ln_eps
    a.ln_colsum, a.rln_gamma, a.rln_beta = _p(ln_colsum), _p(rln_gamma), _p(rln_beta)
    a.incr_counter, a.incr_table = _p(incr_counter), _p(incr_table)
    a.tile = tile
    a.batch2, a.a_batch_stride2, a.out_batch_stride2 = batch2, a_bs2, out_bs2      # second batch level: W / bias shared (grouped conv over clips)
    a.ksplit, a.ksplit_stride = ksplit, ksplit_stride      # S K-slices -> S fp32 partial planes of out_f32 (summed by layernorm(x_planes=S))
    if sched is not None:      # fused scheduler update in the epilogue (resid = x_t, out_f32 = x_{t-1})
        a.sched_fuse, a.sched = 1, sched
    check(lib().fdm_op_gemm(C.byref(a), stream()))


def kv_pad(L):
    """Padded key count of the fragment-packed K / V buffers (whole 32-key tiles)."""
    return (L + 31) // 32 * 32


def kv_buffers(B, H, L, hd, dtype, device):
    """Zeroed fragment-packed K and V buffers ([B*H, Lpad*hd] each; pad keys must stay finite)."""
    Lpad = kv_pad(L)
    return (torch.zeros(B * H, Lpad * hd, device=device, dtype=dtype),
            torch.zeros(B * H, Lpad * hd, device=device, dtype=dtype), Lpad)


def pack_kv(K, V, Kp, Vp, *, B, H, L, Lpad, hd, ldk, ldv):
    check(lib().fdm_op_pack_kv(_p(K), ldk, _p(V), ldv, _p(Kp), _p(Vp), B, H, L, Lpad, hd, code_of(K), stream()))


def attention(Q, Kp, Vp, O, *, B, H, L, hd, ldq, ldo, Lpad, scale, causal=False, slopes=None, period=1):
    a = AttnArgs()
    a.Q, a.ldq, a.Kp, a.Vp, a.Lpad = _p(Q), ldq, _p(Kp), _p(Vp), Lpad
    a.O, a.ldo, a.B, a.H, a.L, a.hd, a.dtype = _p(O), ldo, B, H, L, hd, code_of(Q)
    if isinstance(Q, Split):       # split attention: Q, K, V and O are fp16 plane pairs
        a.q_lo_off, a.kv_lo_off, a.o_lo_off = Q.lo_off, Kp.lo_off, O.lo_off
    elif isinstance(O, Split):     # fp32 attention writing the next GEMM's split operand
        a.o_split, a.o_lo_off = O.code, O.lo_off
    a.scale, a.causal, a.slopes, a.period = scale, int(causal), _p(slopes), period
    check(lib().fdm_op_attention(C.byref(a), stream()))


def layernorm(x, gamma, beta, M, d, *, add_mat=None, add_tab=None, tab_index=None, tab_step=None, eps=1e-5,
              act=ACT_NONE, y_f32=None, y_t=None, dtype=F32, gamma2=None, beta2=None, x_planes=0, x_plane_stride=0):
    a = LnArgs()
    a.x, a.M, a.d, a.add_mat, a.add_tab = _p(x), M, d, _p(add_mat), _p(add_tab)
    a.tab_index, a.tab_step, a.gamma, a.beta, a.eps = _p(tab_index), _p(tab_step), _p(gamma), _p(beta), eps
    a.act, a.y_f32, a.y_t, a.dtype = act, _p(y_f32), _p(y_t), dtype
    a.gamma2, a.beta2, a.y_t_lo_off = _p(gamma2), _p(beta2), _lo(y_t)
    a.x_planes, a.x_plane_stride = x_planes, x_plane_stride
    check(lib().fdm_op_layernorm(C.byref(a), stream()))


def sched_args(mode, x0, x, x_out, n, *, x0u=None, cfg_scale=0.0, n_per_clip=0, tseq=None, step=None, advance=0,
               c1=None, c2=None, sigma=None, sra=None, srm1=None, sqrt_an=None, c_n=None, noise=None, noise_stride=0,
               seed=0, clip0=0, x_out_t=None, arrive=None):
    """fdm_sched_args for fdm_op_sched_step, or (with x0 / x / x_out None) for gemm(..., sched=...)."""
    a = SchedArgs()
    a.x0, a.x0u, a.cfg_scale, a.x, a.x_out = _p(x0), _p(x0u), cfg_scale, _p(x), _p(x_out)
    a.n, a.n_per_clip, a.tseq, a.step, a.advance = n, n_per_clip, _p(tseq), _p(step), advance
    a.c1, a.c2, a.sigma, a.sra, a.srm1 = _p(c1), _p(c2), _p(sigma), _p(sra), _p(srm1)
    a.sqrt_an, a.c_n, a.noise, a.noise_stride = _p(sqrt_an), _p(c_n), _p(noise), noise_stride
    a.seed, a.clip0, a.mode = seed, clip0, mode
    a.x_out_t, a.out_dtype, a.arrive = _p(x_out_t), (code_of(x_out_t) if x_out_t is not None else 0), _p(arrive)
    a.x_out_t_lo_off = _lo(x_out_t)
    return a


def sched_step(mode, x0, x, x_out, n, **kw):
    a = sched_args(mode, x0, x, x_out, n, **kw)
    check(lib().fdm_op_sched_step(C.byref(a), stream()))


def cast(src, dst):
    check(lib().fdm_op_cast(_p(src), _p(dst), src.numel(), code_of(dst), stream()))


def to_operand(src_f32, dtype):
    """fp32 device matrix -> operand-kind copy (identity for fp32; a Split plane pair for the split kinds)."""
    if dtype == F32:
        return src_f32
    if is_split(dtype):
        dst = Split(torch.empty((2,) + tuple(src_f32.shape), dtype=tdtype(dtype), device=src_f32.device), dtype)
    else:
        dst = torch.empty(src_f32.shape, dtype=tdtype(dtype), device=src_f32.device)
    cast(src_f32.contiguous(), dst)
    return dst


def bias_act(inp, vec, out, rows, d, act):
    check(lib().fdm_op_bias_act(_p(inp), _p(vec), _p(out), rows, d, act, stream()))


def add_rows(out, M, d, a, a_div, a_mod, b=None, b_div=1, b_mod=1, c=None, c_div=1, c_mod=1):
    check(lib().fdm_op_add_rows(_p(a), a_div, a_mod, _p(b), b_div, b_mod, _p(c), c_div, c_mod, _p(out), M, d, stream()))


def small_linear(x, W, bias, out, B, K, d, act=ACT_NONE):
    check(lib().fdm_op_small_linear(_p(x), _p(W), _p(bias), _p(out), B, K, d, act, stream()))


def pad_rows(inp, out, B, L, d, pad, zero=False):
    check(lib().fdm_op_pad_rows(_p(inp), _p(out), B, L, d, pad, code_of(inp), int(zero), stream()))


def group_pad(inp, out, B, T, d, groups, pad):
    check(lib().fdm_op_group_pad(_p(inp), _p(out), B, T, d, groups, pad, code_of(inp), stream()))


def conv0(wav, w, bias, out, B, n, T0):
    check(lib().fdm_op_conv0(_p(wav), _p(w), _p(bias), _p(out), B, n, T0, stream()))


def conv0_ln_gelu(wav, w, bias, gamma, beta, out, B, n, T0, eps=1e-5):
    """conv layer 0 + LayerNorm(512) + GELU(erf) of HuBERT-large in one kernel; out [B, T0, 512] fp32 or bf16."""
    check(lib().fdm_op_conv0_ln_gelu(_p(wav), _p(w), _p(bias), _p(gamma), _p(beta), _p(out), _lo(out), B, n, T0, eps, code_of(out), stream()))


def leaky_instnorm(x, B, L, d, *, y_f32=None, y_t=None, eps=1e-5, dtype=F32):
    check(lib().fdm_op_leaky_instnorm(_p(x), _p(y_f32), _p(y_t), B, L, d, eps, dtype, stream()))


def time_groupnorm(x, gamma, beta, B, T, C, *, y_f32=None, y_t=None, eps=1e-5, act=ACT_NONE, dtype=F32, scratch=None):
    """scratch: an 8-byte aligned device buffer (>= B * min(64, ceil(T / 1024)) * C * 16 bytes) lets clips of >= 4096 frames run over
    time chunks (two launches, hundreds of workgroups) instead of C / 64 workgroups per clip."""
    nbytes = scratch.numel() * scratch.element_size() if scratch is not None else 0
    check(lib().fdm_op_time_groupnorm(_p(x), _p(gamma), _p(beta), _p(y_f32), _p(y_t), _lo(y_t), B, T, C, eps, act, dtype, _p(scratch), nbytes, stream()))


def mean_diff(a, b, l1=False):
    """mean((a - b)^2) or mean(|a - b|) as a 1-element device tensor."""
    partial = torch.empty(1024, device=a.device)
    out = torch.empty(1, device=a.device)
    check(lib().fdm_op_mean_diff(_p(a), _p(b), _p(partial), _p(out), a.numel(), int(l1), stream()))
    return out


def linear_interp(x, y, B, Tin, Tout, C):
    check(lib().fdm_op_linear_interp(_p(x), _p(y), B, Tin, Tout, C, stream()))


def vertex_err(gt, pred, region, R, F, V, frame_max, frame_sum, out):
    check(lib().fdm_op_vertex_err(_p(gt), _p(pred), _p(region), R, F, V, _p(frame_max), _p(frame_sum), _p(out), stream()))


def motion_std(verts, tmpl, region, R, F, V, partial, out):
    check(lib().fdm_op_motion_std(_p(verts), _p(tmpl), _p(region), R, F, V, _p(partial), _p(out), stream()))


def adain(content, style, out, NC, Lc, Ls, eps=1e-5):
    check(lib().fdm_op_adain(_p(content), _p(style), _p(out), NC, Lc, Ls, eps, stream()))


def vq_quant(z, codebook, book, B, R, c, K, zq_bcl, idx):
    check(lib().fdm_op_vq_quant(_p(z), _p(codebook), _p(book), B, R, c, K, _p(zq_bcl), _p(idx), stream()))


class Program:
    """A recorded sequence of fdm_op_* launches, replayable as a hipGraph (fdm_prog_*)."""

    def __init__(self):
        h = C.c_void_p()
        check(lib().fdm_prog_create(C.byref(h)))
        self.h = h
        self.keep = []          # tensors referenced by raw pointer inside the program

    def __enter__(self):
        check(lib().fdm_prog_begin(self.h))
        return self

    def __exit__(self, et, ev, tb):
        rc = lib().fdm_prog_end(self.h)
        if et is None:
            check(rc)
        return False

    def hold(self, *tensors):
        self.keep.extend(tensors)

    @property
    def num_ops(self):
        return lib().fdm_prog_num_ops(self.h)

    def run(self):
        check(lib().fdm_prog_run(self.h, stream()))

    def instantiate(self):
        check(lib().fdm_prog_instantiate(self.h, stream()))

    def replay(self, n=1):
        check(lib().fdm_prog_replay(self.h, n, stream()))

    def __del__(self):
        try:
            if self.h:
                lib().fdm_prog_destroy(self.h)
                self.h = None
        except Exception:
            pass
