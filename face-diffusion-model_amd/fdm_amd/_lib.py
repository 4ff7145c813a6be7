"""ctypes binding of libfdm_hip.so (include/fdm_hip.h).  The product path has no CPU fallback:
if the library is missing or a call fails, this module raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FDM_LIB_PATH: load another build of the same library (the AddressSanitizer host build, csrc `make asan`)
LIB_PATH = os.environ.get("FDM_LIB_PATH") or os.path.join(_HERE, "libfdm_hip.so")

F32, BF16, F16X3, F16 = 0, 1, 2, 3      # include/fdm_hip.h FDM_*: operand kinds (F16X3 is a split plane pair; F16 its hi plane alone: denoiser only)
DTYPE_NAMES = {"f32": F32, "bf16": BF16, "f16x3": F16X3, "f16": F16}
ACT_NONE, ACT_RELU, ACT_MISH, ACT_GELU_ERF, ACT_GELU_TANH, ACT_LEAKY02 = range(6)

vp, ll, ci, cf = C.c_void_p, C.c_longlong, C.c_int, C.c_float


class SchedArgs(C.Structure):
    _fields_ = [("x0", vp), ("x0u", vp), ("cfg_scale", cf), ("x", vp), ("x_out", vp),
                ("n", ll), ("n_per_clip", ll), ("tseq", vp), ("step", vp), ("advance", ci),
                ("c1", vp), ("c2", vp), ("sigma", vp), ("sra", vp), ("srm1", vp),
                ("sqrt_an", vp), ("c_n", vp), ("noise", vp), ("noise_stride", ll), ("x_out_t", vp), ("out_dtype", ci), ("arrive", vp),
                ("seed", C.c_ulonglong), ("clip0", ci),
                ("mode", ci), ("x_out_t_lo_off", ll), ("seed_dev", vp)]


class GemmArgs(C.Structure):
    _fields_ = [("A", vp), ("lda", ll), ("a_batch_stride", ll),
                ("W", vp), ("ldw", ll), ("w_batch_stride", ll),
                ("M", ci), ("N", ci), ("K", ci), ("batch", ci), ("dtype", ci),
                ("bias", vp), ("bias_batch_stride", ll), ("act", ci),
                ("resid", vp), ("ldr", ll), ("resid_row_mod", ci),
                ("out_f32", vp), ("ldo_f32", ll), ("out_t", vp), ("ldo_t", ll),
                ("out_batch_stride", ll),
                ("out_kp", vp), ("kp_col0", ci), ("out_vp", vp), ("vp_col0", ci),
                ("kv_L", ci), ("kv_Lpad", ci), ("kv_hd", ci),
                ("stat_out", vp), ("ln_stat_in", vp), ("ln_nparts", ci), ("ln_dim", ci), ("ln_eps", cf),
                ("ln_colsum", vp), ("rln_gamma", vp), ("rln_beta", vp), ("incr_counter", vp), ("incr_table", vp), ("tile", ci),
                ("sched_fuse", ci), ("sched", SchedArgs),
                ("a_lo_off", ll), ("w_lo_off", ll), ("out_t_lo_off", ll), ("kv_lo_off", ll),
                ("ksplit", ci), ("ksplit_stride", ll), ("batch2", ci), ("a_batch_stride2", ll), ("out_batch_stride2", ll)]


# the eight tiles of include/fdm_hip.h, and the retired ids (accepted: they resolve to a live tile)
TILE_AUTO, TILE_64x64, TILE_128x64, TILE_128x128, TILE_64x64_S2, TILE_32x64_S3, TILE_256x128_PP, TILE_80x128, TILE_64x128 = 0, 1, 2, 3, 8, 9, 10, 11, 12
TILE_96x128, TILE_256x128, TILE_64x64_S3, TILE_128x64_S3 = 4, 5, 6, 7


class ModelDesc(C.Structure):
    _fields_ = [(n, ci) for n in ("d", "n_head", "n_layers", "ffn", "G", "c", "n_style", "n_emo", "audio_in", "pair",
                                  "pe_periodic", "period", "latent_mish", "style_mish", "max_len")]


class SampleArgs(C.Structure):
    _fields_ = [("kind", ci), ("x_T", vp), ("out", vp), ("t_list", vp), ("n_steps", ci), ("ddim_steps", ci), ("noise", vp),
                ("seed", C.c_ulonglong), ("clip0", ci), ("cfg_scale", cf), ("eager", ci), ("record", vp), ("graph_steps", ci)]


class VqDesc(C.Structure):
    _fields_ = [(n, ci) for n in ("G", "c", "K", "n_books", "V3", "pre")]


class AttnArgs(C.Structure):
    _fields_ = [("Q", vp), ("ldq", ll), ("Kp", vp), ("Vp", vp), ("Lpad", ci),
                ("O", vp), ("ldo", ll), ("B", ci), ("H", ci), ("L", ci), ("hd", ci), ("dtype", ci),
                ("scale", cf), ("causal", ci), ("slopes", vp), ("period", ci), ("o_split", ci), ("o_lo_off", ll), ("q_lo_off", ll), ("kv_lo_off", ll)]


class LnArgs(C.Structure):
    _fields_ = [("x", vp), ("M", ci), ("d", ci), ("add_mat", vp), ("add_tab", vp),
                ("tab_index", vp), ("tab_step", vp), ("gamma", vp), ("beta", vp), ("eps", cf),
                ("act", ci), ("y_f32", vp), ("y_t", vp), ("dtype", ci), ("gamma2", vp), ("beta2", vp), ("y_t_lo_off", ll),
                ("add_mat_L", ci), ("add_mat_group", ci), ("add_mat_wrap", ci), ("x_planes", ci), ("x_plane_stride", ll)]


# public structs of include/fdm_hip.h -> their mirrors (sizes checked against the loaded library in lib())
STRUCTS = {"fdm_sched_args": SchedArgs, "fdm_gemm_args": GemmArgs, "fdm_attn_args": AttnArgs, "fdm_ln_args": LnArgs,
           "fdm_model_desc": ModelDesc, "fdm_sample_args": SampleArgs, "fdm_vq_desc": VqDesc}

# every symbol include/fdm_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "fdm_last_error": (C.c_char_p, []),
    "fdm_version": (ci, []),
    "fdm_abi_struct_size": (ci, [C.c_char_p]),
    "fdm_device_ok": (ci, []),
    "fdm_op_gemm": (ci, [C.POINTER(GemmArgs), vp]),
    "fdm_gemm_heuristic_tile": (ci, [C.POINTER(GemmArgs)]),
    "fdm_op_attention": (ci, [C.POINTER(AttnArgs), vp]),
    "fdm_op_pack_kv": (ci, [vp, ll, vp, ll, vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    "fdm_op_layernorm": (ci, [C.POINTER(LnArgs), vp]),
    "fdm_op_sched_step": (ci, [C.POINTER(SchedArgs), vp]),
    "fdm_op_cast": (ci, [vp, vp, ll, ci, vp]),
    "fdm_op_vertex_err": (ci, [vp, vp, vp, ci, ci, ci, vp, vp, vp, vp]),
    "fdm_op_motion_std": (ci, [vp, vp, vp, ci, ci, ci, vp, vp, vp]),
    "fdm_op_linear_interp": (ci, [vp, vp, ci, ci, ci, ci, vp]),
    "fdm_op_bias_act": (ci, [vp, vp, vp, ll, ci, ci, vp]),
    "fdm_op_add_rows": (ci, [vp, ci, ci, vp, ci, ci, vp, ci, ci, vp, ll, ci, vp]),
    "fdm_op_small_linear": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp]),
    "fdm_op_pad_rows": (ci, [vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    "fdm_op_group_pad": (ci, [vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    "fdm_op_conv0": (ci, [vp, vp, vp, vp, ci, ci, ci, vp]),
    "fdm_op_conv0_ln_gelu": (ci, [vp, vp, vp, vp, vp, vp, ll, ci, ci, ci, cf, ci, vp]),
    "fdm_op_leaky_instnorm": (ci, [vp, vp, vp, ci, ci, ci, cf, ci, vp]),
    "fdm_op_adain": (ci, [vp, vp, vp, ci, ci, ci, cf, vp]),
    "fdm_op_mean_diff": (ci, [vp, vp, vp, vp, ll, ci, vp]),
    "fdm_op_time_groupnorm": (ci, [vp, vp, vp, vp, vp, ll, ci, ci, ci, cf, ci, ci, vp, ll, vp]),
    "fdm_op_vq_quant": (ci, [vp, vp, vp, ci, ci, ci, ci, vp, vp, vp]),
    "fdm_op_vq_stats": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, cf, vp, vp, vp, vp, vp]),
    "fdm_prog_create": (ci, [C.POINTER(vp)]),
    "fdm_prog_destroy": (ci, [vp]),
    "fdm_prog_begin": (ci, [vp]),
    "fdm_prog_end": (ci, [vp]),
    "fdm_prog_run": (ci, [vp, vp]),
    "fdm_prog_instantiate": (ci, [vp, vp]),
    "fdm_prog_replay": (ci, [vp, ci, vp]),
    "fdm_prog_num_ops": (ci, [vp]),
    "fdm_model_preset": (ci, [C.c_char_p, C.POINTER(ModelDesc)]),
    "fdm_plan_create": (ci, [C.POINTER(ModelDesc), ci, ci, ci, ci, C.POINTER(vp)]),
    "fdm_plan_reserve": (ci, [vp, ci, ci, ci]),
    "fdm_plan_destroy": (ci, [vp]),
    "fdm_plan_set_weights": (ci, [vp, C.c_char_p, vp, ll, vp]),
    "fdm_plan_commit": (ci, [vp, vp]),
    "fdm_audio_prepare": (ci, [vp, vp, ci, ci, ci, vp, vp, ci, ci, vp]),
    "fdm_audio_prepare_conds": (ci, [vp, vp, ci, ci, ci, ci, vp, vp, ci, ci, vp]),
    "fdm_denoise_step": (ci, [vp, vp, ci, cf, vp, vp, vp]),
    "fdm_sample_graph": (ci, [vp, C.POINTER(SampleArgs), vp]),
    "fdm_plan_tune": (ci, [vp, vp]),
    "fdm_plan_get": (ci, [vp, C.c_char_p, C.POINTER(ll)]),
    "fdm_plan_set": (ci, [vp, C.c_char_p, ll]),
    "fdm_hubert_create": (ci, [ci, ci, ci, C.POINTER(vp)]),
    "fdm_hubert_set_weights": (ci, [vp, C.c_char_p, vp, ll, vp]),
    "fdm_hubert_forward": (ci, [vp, vp, ci, ci, ci, ci, ci, vp, C.POINTER(ci), vp]),
    "fdm_hubert_frames": (ci, [ci]),
    "fdm_hubert_destroy": (ci, [vp]),
    "fdm_vq_create": (ci, [C.POINTER(VqDesc), ci, C.POINTER(vp)]),
    "fdm_vq_set_weights": (ci, [vp, C.c_char_p, vp, ll, vp]),
    "fdm_vq_quant": (ci, [vp, vp, vp, ci, ci, vp, vp, vp]),
    "fdm_vq_quant_stats": (ci, [vp, vp, vp, vp, ci, ci, cf, vp, vp, vp]),
    "fdm_vq_decode": (ci, [vp, vp, ci, ci, vp, vp]),
    "fdm_vq_encode": (ci, [vp, vp, vp, ci, ci, vp, vp]),
    "fdm_vq_destroy": (ci, [vp]),
    "fdm_schedule_host": (ci, [ci, vp]),
    "fdm_ddim_schedule_host": (ci, [ci, ci, vp, vp, vp, vp]),
    "fdm_alibi_slopes_host": (ci, [ci, vp]),
    "fdm_pe_table_host": (ci, [ci, ci, ci, ci, vp]),
}

_lib = None
LIB_VERSION = 105      # include/fdm_hip.h as this binding mirrors it (fdm_version()): struct layouts AND entry-point signatures


class FdmError(RuntimeError):
    pass


def lib():
    """Load (once) and return the bound library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FdmError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the product path)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        # a signature changed in place under an old symbol name would corrupt arguments silently: the build must be the one this
        # binding was written for
        if l.fdm_version() != LIB_VERSION:
            raise FdmError(f"{LIB_PATH}: fdm_version() = {l.fdm_version()}, this binding is for {LIB_VERSION}: rebuild the library "
                           "(python -c 'import __graft_entry__ as g; g.build()')")
        # the ctypes mirrors above must be the structs this build was compiled with (a stale .so or a header edit that
        # missed this file would otherwise corrupt arguments silently)
        for cname, mirror in STRUCTS.items():
            n = l.fdm_abi_struct_size(cname.encode())
            if n != C.sizeof(mirror):
                raise FdmError(f"{LIB_PATH}: sizeof({cname}) = {n} in the library, {C.sizeof(mirror)} in fdm_amd/_lib.py: rebuild the "
                               "library (python -c 'import __graft_entry__ as g; g.build()') or update the binding")
        _lib = l
    return _lib


def check(code):
    if code != 0:
        raise FdmError(f"libfdm_hip error {code}: {lib().fdm_last_error().decode()}")
