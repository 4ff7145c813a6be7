"""CPU: the C-ABI library loads and exports every symbol include/fdm_hip.h declares
(no compute call without a GPU); host-side argument validation reports errors, not crashes."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "fdm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fdm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from fdm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    l = _lib.lib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(l, n), f"{n} declared in include/fdm_hip.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes binding"
    assert l.fdm_version() >= 103
    import ctypes as _C
    for cname, mirror in _lib.STRUCTS.items():          # the ctypes mirrors are the structs the library was built with
        assert l.fdm_abi_struct_size(cname.encode()) == _C.sizeof(mirror), cname
    assert l.fdm_abi_struct_size(b"no_such_struct") == -1


def test_argument_validation_without_device():
    from fdm_amd import _lib
    l = _lib.lib()
    a = _lib.GemmArgs()
    assert l.fdm_op_gemm(C.byref(a), None) == -1
    assert b"null operand" in l.fdm_last_error()
    a.A, a.W, a.M, a.N, a.K, a.dtype, a.lda, a.ldw = 16, 16, 4, 4, 48, 0, 48, 48
    assert l.fdm_op_gemm(C.byref(a), None) == -2
    assert b"multiple of 32" in l.fdm_last_error()
    s = _lib.SchedArgs()
    s.x0, s.x_out, s.n = 16, 16, 6
    assert l.fdm_op_sched_step(C.byref(s), None) == -2
    at = _lib.AttnArgs()
    at.Q = at.Kp = at.Vp = at.O = 16
    at.hd = 96
    assert l.fdm_op_attention(C.byref(at), None) == -2


def test_argument_validation_of_the_round1_additions():
    """pack_kv / packed K,V outputs / tile / metrics / interpolation: bad arguments are reported, nothing is launched."""
    from fdm_amd import _lib
    l = _lib.lib()
    assert l.fdm_op_pack_kv(16, 64, 16, 64, 16, 16, 1, 1, 10, 31, 64, _lib.BF16, None) == -2       # Lpad % 32
    assert l.fdm_op_pack_kv(None, 64, 16, 64, 16, 16, 1, 1, 10, 32, 64, _lib.BF16, None) == -1
    a = _lib.GemmArgs()
    a.A, a.W, a.M, a.N, a.K, a.dtype, a.lda, a.ldw = 16, 16, 64, 192, 64, _lib.BF16, 64, 64
    a.out_t, a.out_kp, a.kp_col0, a.out_vp, a.vp_col0 = 16, 16, 64, 16, 128
    a.kv_L, a.kv_Lpad, a.kv_hd = 32, 32, 24                                                       # hd % 16
    assert l.fdm_op_gemm(C.byref(a), None) == -2 and b"packed K/V" in l.fdm_last_error()
    a.kv_hd, a.kv_L = 64, 48                                                                      # M % L
    assert l.fdm_op_gemm(C.byref(a), None) == -2
    a.out_kp = a.out_vp = None
    a.tile = 19
    assert l.fdm_op_gemm(C.byref(a), None) == -1 and b"tile" in l.fdm_last_error()
    assert l.fdm_op_vertex_err(16, 16, None, 5, 3, 10, 16, 16, 16, None) == -2                    # region NULL needs R == V
    assert l.fdm_op_motion_std(16, None, None, 10, 3, 10, 16, 16, None) == -1
    assert l.fdm_op_linear_interp(16, 16, 1, 0, 5, 4, None) == -1


def test_argument_validation_of_the_round5_additions():
    """K slices (fdm_gemm_args.ksplit / fdm_ln_args.x_planes), the second batch level (batch2), the lockstep flag and the retired tile
    ids: bad arguments are reported before anything is launched; retired ids and the lockstep flag are accepted (recorded here)."""
    from fdm_amd import _lib
    l = _lib.lib()

    def args():
        a = _lib.GemmArgs()
        a.A, a.W, a.M, a.N, a.K, a.dtype, a.lda, a.ldw, a.out_f32, a.ldo_f32, a.ldr, a.ldo_t = 16, 16, 64, 128, 1024, _lib.BF16, 1024, 1024, 16, 128, 128, 128
        a.batch = 1
        return a
    a = args(); a.ksplit = 5
    assert l.fdm_op_gemm(C.byref(a), None) == -1 and b"ksplit" in l.fdm_last_error()
    a = args(); a.ksplit, a.ksplit_stride = 3, 64 * 128                     # 16 k-tiles are not divisible by 3
    assert l.fdm_op_gemm(C.byref(a), None) == -2 and b"does not divide" in l.fdm_last_error()
    a = args(); a.ksplit, a.ksplit_stride, a.act = 2, 64 * 128, _lib.ACT_RELU
    assert l.fdm_op_gemm(C.byref(a), None) == -1 and b"plain launch" in l.fdm_last_error()
    a = args(); a.ksplit, a.ksplit_stride = 2, 100                           # shorter than one output plane
    assert l.fdm_op_gemm(C.byref(a), None) == -2
    a = args(); a.ksplit, a.ksplit_stride, a.tile = 2, 64 * 128, _lib.TILE_128x128
    assert l.fdm_op_gemm(C.byref(a), None) == -1 and b"64-column tiles" in l.fdm_last_error()
    a = args(); a.batch2, a.ksplit, a.ksplit_stride = 2, 2, 64 * 128
    assert l.fdm_op_gemm(C.byref(a), None) == -1 and b"batch2" in l.fdm_last_error()
    a = args(); a.batch2, a.tile = 2, _lib.TILE_80x128
    assert l.fdm_op_gemm(C.byref(a), None) == -1 and b"batch2 runs on" in l.fdm_last_error()
    a = args(); a.lda = 1 << 31                                              # row strides reach the kernel as 32-bit preloaded arguments
    assert l.fdm_op_gemm(C.byref(a), None) == -2 and b"32-bit" in l.fdm_last_error()
    ln = _lib.LnArgs()
    ln.x, ln.gamma, ln.beta, ln.M, ln.d, ln.y_f32, ln.dtype = 16, 16, 16, 4, 256, 16, _lib.F32
    ln.x_planes, ln.x_plane_stride = 5, 4096
    assert l.fdm_op_layernorm(C.byref(ln), None) == -1 and b"x_planes" in l.fdm_last_error()
    ln.x_planes, ln.x_plane_stride = 2, 512                                  # < M * d
    assert l.fdm_op_layernorm(C.byref(ln), None) == -1
    # accepted forms, recorded (nothing is launched on this machine)
    h = C.c_void_p()
    assert l.fdm_prog_create(C.byref(h)) == 0 and l.fdm_prog_begin(h) == 0
    for tile in (_lib.TILE_64x64 | 0x200, _lib.TILE_64x64_S3, _lib.TILE_128x64_S3, _lib.TILE_256x128, _lib.TILE_96x128 | 0x100):
        a = args(); a.tile = tile
        assert l.fdm_op_gemm(C.byref(a), None) == 0, tile
    a = args(); a.ksplit, a.ksplit_stride, a.tile = 4, 64 * 128, _lib.TILE_32x64_S3
    assert l.fdm_op_gemm(C.byref(a), None) == 0
    a = args(); a.batch, a.batch2, a.tile = 16, 3, _lib.TILE_128x64
    assert l.fdm_op_gemm(C.byref(a), None) == 0
    ln.x_planes, ln.x_plane_stride = 4, 1024
    assert l.fdm_op_layernorm(C.byref(ln), None) == 0
    assert l.fdm_prog_end(h) == 0 and l.fdm_prog_num_ops(h) == 8 and l.fdm_prog_destroy(h) == 0
    assert l.fdm_version() == _lib.LIB_VERSION == 105
    import ctypes
    for sym in ("fdm_prog_set_lane", "fdm_prog_run_lanes"):                 # removed this round
        with pytest.raises(AttributeError):
            getattr(ctypes.CDLL(_lib.LIB_PATH), sym)


def test_product_path_has_no_cpu_fallback():
    import torch
    from fdm_amd import _lib, ops
    with pytest.raises(_lib.FdmError):
        ops.cast(torch.zeros(8), torch.zeros(8))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "face-diffusion-model_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py") or f.endswith(".hip") or f.endswith(".hpp"):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dp, f)


def test_program_recording_and_split_validation_without_device():
    """fdm_prog_* records validated launches without touching a device: record, count, destroy (also the part of the host
    layer the AddressSanitizer build exercises, tools/run_asan_tests.sh); split-operand arguments are validated."""
    from fdm_amd import _lib
    l = _lib.lib()
    h = C.c_void_p()
    assert l.fdm_prog_create(C.byref(h)) == 0
    assert l.fdm_prog_begin(h) == 0
    assert l.fdm_prog_begin(h) == -4                       # already recording on this thread
    a = _lib.GemmArgs()
    a.A, a.W, a.M, a.N, a.K, a.dtype, a.lda, a.ldw, a.out_f32, a.ldo_f32, a.ldr, a.ldo_t = 16, 16, 64, 64, 64, _lib.BF16, 64, 64, 16, 64, 64, 64
    for _ in range(5):
        assert l.fdm_op_gemm(C.byref(a), None) == 0        # recorded, not launched
    a.dtype = _lib.F16X3
    assert l.fdm_op_gemm(C.byref(a), None) == -1 and b"a_lo_off" in l.fdm_last_error()
    a.a_lo_off = a.w_lo_off = 4096
    a.out_t = 16
    assert l.fdm_op_gemm(C.byref(a), None) == -1 and b"out_t_lo_off" in l.fdm_last_error()
    a.out_t_lo_off = 4096
    assert l.fdm_op_gemm(C.byref(a), None) == 0
    assert l.fdm_op_cast(16, 16, 128, _lib.F16X3, None) == 0
    assert l.fdm_op_cast(16, 16, 128, 7, None) == -1
    ln = _lib.LnArgs()
    ln.x, ln.gamma, ln.beta, ln.M, ln.d, ln.y_t, ln.dtype = 16, 16, 16, 4, 256, 16, _lib.F16X3
    assert l.fdm_op_layernorm(C.byref(ln), None) == -1 and b"y_t_lo_off" in l.fdm_last_error()
    at = _lib.AttnArgs()
    at.Q = at.Kp = at.Vp = at.O = 16
    at.hd, at.B, at.H, at.L, at.Lpad, at.ldq, at.ldo, at.dtype, at.o_split = 64, 1, 1, 8, 32, 64, 64, _lib.BF16, _lib.F16X3
    assert l.fdm_op_attention(C.byref(at), None) == -1 and b"o_split" in l.fdm_last_error()
    assert l.fdm_prog_end(h) == 0
    assert l.fdm_prog_num_ops(h) == 7
    assert l.fdm_prog_replay(h, 1, None) == -4             # not instantiated
    assert l.fdm_prog_destroy(h) == 0
    assert l.fdm_plan_get(None, b"rows", C.byref(C.c_longlong())) == -1


def test_plan_layer_argument_validation_without_device():
    """The plan layer reports bad arguments before touching a device (null handles, bad geometry / kinds / dtypes)."""
    from fdm_amd import _lib
    l = _lib.lib()
    assert l.fdm_plan_reserve(None, 1, 10, 0) == -1
    assert l.fdm_plan_commit(None, None) == -1
    assert l.fdm_plan_set_weights(None, b"x", 16, 4, None) == -1
    assert l.fdm_audio_prepare(None, 16, 1, 10, 1024, 16, None, 10, 0, None) == -1
    assert l.fdm_audio_prepare_conds(None, 16, 1, 10, 1024, 8, 16, None, 10, 0, None) == -1
    assert l.fdm_denoise_step(None, 16, 0, 0.0, 16, None, None) == -1
    assert l.fdm_sample_graph(None, C.byref(_lib.SampleArgs()), None) == -1
    assert l.fdm_plan_tune(None, None) == -1
    assert l.fdm_plan_destroy(None) == 0
    h = C.c_void_p()
    assert l.fdm_hubert_create(2, 0, 0, C.byref(h)) == -1 and b"kind" in l.fdm_last_error()
    assert l.fdm_hubert_create(0, 0, 3, C.byref(h)) == -1                      # audio encoders: fp32 / bf16 / f16x3 (split-fp16 layers)
    assert l.fdm_hubert_create(0, 0, _lib.F16X3, C.byref(h)) == 0 and l.fdm_hubert_destroy(h) == 0
    assert l.fdm_hubert_create(1, 0, 0, C.byref(h)) == 0                                 # creation itself needs no device
    assert l.fdm_hubert_set_weights(h, b"encoder.layer_norm.weight", None, 768, None) == -1
    assert l.fdm_hubert_destroy(h) == 0
    assert l.fdm_hubert_frames(32000) == 98 and l.fdm_hubert_frames(32080) == 100 and l.fdm_hubert_frames(160000) == 498
    assert l.fdm_hubert_frames(100) == 0
    d = _lib.VqDesc(16, 64, 256, 1, 15069, 0)
    v = C.c_void_p()
    assert l.fdm_vq_create(C.byref(d), 0, C.byref(v)) == 0 and l.fdm_vq_destroy(v) == 0
    d.c = 200
    assert l.fdm_vq_create(C.byref(d), 0, C.byref(v)) == -2
    d = _lib.VqDesc(8, 64, 256, 7, 15069, 0)                                             # G * c != 1024 without a pre-embedding
    assert l.fdm_vq_create(C.byref(d), 0, C.byref(v)) == -2
    pd = _lib.ModelDesc()
    assert l.fdm_model_preset(b"biwi", C.byref(pd)) == 0 and pd.d // pd.n_head == 256
    p = C.c_void_p()
    # geometry is checked before the device: head_dim 256 is valid in every mode (FDM_F16X3 runs the one-wave-per-SIMD form of the split attention kernel there), so on
    # this box the call gets as far as "no device"; an unsupported head_dim is a shape error
    rc = l.fdm_plan_create(C.byref(pd), 1, 10, 0, _lib.F16X3, C.byref(p))
    assert (rc == -4 and b"no gfx950 device" in l.fdm_last_error()) or (rc == 0 and l.fdm_plan_destroy(p) == 0)
    # FDM_F16 (round 6: single-plane fp16) is a kind of the step program only: the plan takes it, the audio encoders (above) and the VQ stages do not
    rc = l.fdm_plan_create(C.byref(pd), 1, 10, 0, _lib.F16, C.byref(p))
    assert (rc == -4 and b"no gfx950 device" in l.fdm_last_error()) or (rc == 0 and l.fdm_plan_destroy(p) == 0)
    assert l.fdm_plan_create(C.byref(pd), 1, 10, 0, 4, C.byref(p)) == -1 and b"dtype" in l.fdm_last_error()
    dq = _lib.VqDesc(16, 64, 256, 1, 15069, 0)
    assert l.fdm_vq_create(C.byref(dq), _lib.F16, C.byref(v)) == -1
    pd.n_head = 32
    assert l.fdm_plan_create(C.byref(pd), 1, 10, 0, _lib.F16X3, C.byref(p)) == -2 and b"head_dim" in l.fdm_last_error()


def test_tile_heuristic_is_a_pure_function_of_the_shape():
    """fdm_gemm_heuristic_tile (host only): the FDM_TILE_* a launch with tile = 0 resolves to.  Pins the rules the round-3 sweep
    derived (profiles/r3_tile_sweep/): one-round grids (80x128 at 800 x 3072, 128x128 at 1992 x 2048, 256x128 at 2400 x 3072),
    128x64 rings above 1024 rows, 64x64 for the step's N = 1024 sites at 800 rows, short clips in the split kind on 32-row tiles;
    the fp32 kind keeps the round-2 thresholds."""
    from fdm_amd import _lib
    l = _lib.lib()

    def tile(dtype, M, N, K, batch=0, sched_fuse=0):
        a = _lib.GemmArgs()
        a.M, a.N, a.K, a.dtype, a.batch, a.sched_fuse = M, N, K, dtype, batch, sched_fuse
        a.tile = 7                      # ignored by the query
        return l.fdm_gemm_heuristic_tile(C.byref(a))

    assert l.fdm_gemm_heuristic_tile(None) == -1
    B, F, S = _lib.BF16, _lib.F32, _lib.F16X3
    assert tile(B, 800, 3072, 1024) == _lib.TILE_80x128 and tile(S, 800, 3072, 1024) == _lib.TILE_80x128 and tile(F, 800, 3072, 1024) == _lib.TILE_80x128
    assert tile(B, 800, 1024, 1024) == _lib.TILE_64x64 and tile(B, 800, 1024, 2048) == _lib.TILE_64x64
    assert tile(B, 800, 2048, 1024) == _lib.TILE_64x128 and tile(S, 800, 2048, 1024) == _lib.TILE_64x128 and tile(F, 800, 2048, 1024) == _lib.TILE_64x64      # FFN1: 13 x 16 = 208 tiles, one round
    assert tile(B, 100, 1024, 1024) == _lib.TILE_32x64_S3 and tile(S, 100, 1024, 1024) == _lib.TILE_32x64_S3 and tile(B, 200, 512, 512) == _lib.TILE_32x64_S3      # one short clip: 32-row tiles
    assert tile(S, 600, 3072, 1024) == _lib.TILE_64x128 and tile(S, 400, 3072, 1024) == _lib.TILE_128x64 and tile(B, 400, 3072, 1024) == _lib.TILE_128x64 and tile(S, 800, 1024, 1024) == _lib.TILE_64x64      # 10 x 24 = 240 one-round tiles | 4 x 48 = 192 tiles of 128x64: (nearly) one round
    assert tile(B, 1992, 2048, 1024) == _lib.TILE_128x128 and tile(S, 1992, 2048, 1024) == _lib.TILE_128x128      # 16 x 16 = 256 tiles
    assert tile(B, 2400, 3072, 1024) == _lib.TILE_256x128_PP                                                         # 10 x 24 = 240 tiles
    assert tile(B, 1200, 1024, 1024) == _lib.TILE_128x64 and tile(B, 3000, 1024, 1024) == _lib.TILE_128x64 and tile(B, 3200, 1024, 1024) == _lib.TILE_128x128
    assert tile(S, 1200, 1024, 1024) == _lib.TILE_128x64 and tile(S, 2400, 2048, 1024) == _lib.TILE_80x128      # (beyond the rules: fewest operand rows on the busiest CU)
    assert tile(B, 1200, 512, 512) == _lib.TILE_64x64                   # MEAD's short-K sites stay on the resident 64x64 grid
    assert tile(F, 1992, 2048, 1024) == _lib.TILE_64x64 and tile(F, 1200, 1024, 1024) == _lib.TILE_64x64
    assert tile(B, 6400, 3072, 1024) == _lib.TILE_128x128 and tile(S, 6400, 1024, 2048) == _lib.TILE_128x128
    assert tile(B, 6400, 1024, 2048) == _lib.TILE_256x128_PP and tile(B, 3200, 2048, 1024) == _lib.TILE_256x128_PP      # thousands of rows: the ping-pong loop
    assert tile(B, 6400, 1024, 1024, sched_fuse=1) == _lib.TILE_256x128_PP and tile(F, 6400, 1024, 1024, sched_fuse=1) == _lib.TILE_64x64
    assert tile(B, 2400, 1024, 1024) == _lib.TILE_80x128                # 30 x 8 = 240 tiles
    assert tile(B, 800, 1024, 1024, sched_fuse=1) == _lib.TILE_64x64
    assert tile(B, 100, 1024, 1024, batch=8) == _lib.TILE_64x64
    # round 5 (loader-wave build, profiles/r5_tile_sweep/): a half-empty last round of 128x128 -> 128x64; narrow outputs on one round of 64x128
    assert tile(B, 3200, 3072, 1024) == _lib.TILE_128x64 and tile(B, 2400, 512, 1024) == _lib.TILE_64x128 and tile(S, 800, 1536, 512) == _lib.TILE_64x128
    assert tile(S, 1600, 3072, 1024) == _lib.TILE_80x128 and tile(S, 2400, 3072, 1024) == _lib.TILE_128x128
