"""GPU: every fdm_op_* kernel against a plain fp32 torch (CPU) reference of the same operator,
called through the C ABI.  fp32 kernels: 2e-5 relative-to-scale; bf16 kernels: 2e-2 (bf16 inputs have
8 significant bits; references are computed from the bf16-rounded inputs so only accumulation differs)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from fdm_amd import ops  # noqa: E402
from fdm_amd._lib import (ACT_GELU_ERF, ACT_GELU_TANH, ACT_LEAKY02, ACT_MISH, ACT_NONE, ACT_RELU, BF16, F16, F16X3, F32)  # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def act_ref(x, act):
    if act == ACT_RELU:
        return torch.relu(x)
    if act == ACT_MISH:
        return F.mish(x)
    if act == ACT_GELU_ERF:
        return F.gelu(x)
    if act == ACT_GELU_TANH:
        return x * 0.5 * (1.0 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))
    if act == ACT_LEAKY02:
        return F.leaky_relu(x, 0.2)
    return x


@pytest.mark.parametrize("dtype", [F32, BF16, F16X3])
@pytest.mark.parametrize("M,N,K,S,tile", [(100, 1024, 1024, 4, 9), (249, 512, 1024, 2, 1), (800, 1024, 2048, 4, 8), (100, 1024, 2048, 4, 9),
                                          (37, 256, 512, 2, 9), (800, 1024, 1024, 2, 0)])
def test_gemm_split_k_planes_summed_by_layernorm(dtype, M, N, K, S, tile):
    """fdm_gemm_args.ksplit: S K-slices write S fp32 partial planes (slice 0 with bias + residual), fdm_ln_args.x_planes sums them
    in plane order.  (a) the planes add up to the unsplit product (fp32-class: only the association of the k sum differs);
    (b) every slice equals an ordinary GEMM over its own K range, bit for bit; (c) LayerNorm over the planes == LayerNorm of their
    fixed-order sum, bit for bit; (d) the tile changes nothing."""
    g = torch.Generator().manual_seed(M + N + K + S)
    A32 = torch.randn(M, K, generator=g)
    W32 = torch.randn(N, K, generator=g) / math.sqrt(K)
    A, Wt = ops.to_operand(A32.to(DEV), dtype), ops.to_operand(W32.to(DEV), dtype)
    bias = torch.randn(N, generator=g).to(DEV)
    resid = torch.randn(M, N, generator=g).to(DEV)
    planes = torch.full((S, M, N), float("nan"), device=DEV)
    ops.gemm(A, Wt, M, N, K, bias=bias, resid=resid, out_f32=planes, tile=tile, ksplit=S, ksplit_stride=M * N)
    one = torch.zeros(M, N, device=DEV)
    ops.gemm(A, Wt, M, N, K, bias=bias, resid=resid, out_f32=one)
    torch.cuda.synchronize()
    assert torch.isfinite(planes).all()
    acc = planes[0].clone()
    for s_ in range(1, S):
        acc += planes[s_]
    assert rel(acc, one) < 2e-6      # same operands in every kind: only the association of the k sum differs
    # (b) slice s == the GEMM over columns [s K / S, (s + 1) K / S) of both operands (lda = ldw = K, pointers advanced)
    Ks = K // S
    for s_ in (0, S - 1):
        ref = torch.zeros(M, N, device=DEV)
        ops.gemm(ops.cols(A, s_ * Ks), ops.cols(Wt, s_ * Ks), M, N, Ks, lda=K, ldw=K, bias=bias if s_ == 0 else None, resid=resid if s_ == 0 else None, out_f32=ref)
        torch.cuda.synchronize()
        assert torch.equal(ref, planes[s_]), s_
    # (c) LayerNorm over the planes
    if N in (256, 512, 768, 1024):
        gam, bet = torch.randn(N, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
        y1, y2 = torch.zeros(M, N, device=DEV), torch.zeros(M, N, device=DEV)
        ops.layernorm(planes, gam, bet, M, N, y_f32=y1, x_planes=S, x_plane_stride=M * N)
        ops.layernorm(acc, gam, bet, M, N, y_f32=y2)
        torch.cuda.synchronize()
        assert torch.equal(y1, y2)
    # (d) another tile, same bits
    p2 = torch.zeros(S, M, N, device=DEV)
    ops.gemm(A, Wt, M, N, K, bias=bias, resid=resid, out_f32=p2, tile=(8 if tile != 8 else 1), ksplit=S, ksplit_stride=M * N)
    torch.cuda.synchronize()
    assert torch.equal(p2, planes)


@pytest.mark.parametrize("dtype", [F32, BF16, F16X3])
@pytest.mark.parametrize("C,G,T,dg,KW,tile", [(4, 16, 123, 64, 16, 0), (3, 16, 70, 48, 8, 1), (2, 4, 200, 64, 8, 2), (1, 16, 50, 64, 8, 0)])
def test_gemm_second_batch_level_equals_per_clip_launches(dtype, C, G, T, dg, KW, tile):
    """fdm_gemm_args.batch2: a grouped Conv1d (HuBERT's positional conv: models/hubert.py:110-137 over transformers'
    HubertPositionalConvEmbedding) for all clips in ONE launch -- z = (clip, group), workgroups dealt so that an XCD serves its
    own groups only -- against one launch per clip: bit-identical, in the lean and the edge-handling kernel (dg = 48), for a group
    count that does (16) and does not (4) divide over the 8 XCDs, and == a torch grouped conv."""
    g = torch.Generator().manual_seed(C * 100 + T)
    D = G * dg
    x = torch.randn(C, T + KW, D, generator=g)                                  # padded channels-last signal
    w = torch.randn(G, dg, KW * dg, generator=g) / math.sqrt(KW * dg)            # [group][out][tap * in]
    bias = torch.randn(D, generator=g).to(DEV)
    resid = torch.randn(C * T, D, generator=g).to(DEV)
    xg32 = x.reshape(C, T + KW, G, dg).permute(2, 0, 1, 3).contiguous().to(DEV)   # [G, C, T + KW, dg]
    xg = ops.to_operand(xg32.reshape(G * C * (T + KW), dg), dtype)
    wt = ops.to_operand(w.reshape(G * dg, KW * dg).to(DEV), dtype)
    kw = dict(lda=dg, batch=G, a_bs=C * (T + KW) * dg, w_bs=dg * KW * dg, bias=bias, bias_bs=dg, act=ACT_GELU_ERF, ldr=D, ldo_f32=D, out_bs=dg)
    one = torch.zeros(C * T, D, device=DEV)
    ops.gemm(xg, wt, T, dg, KW * dg, resid=resid, out_f32=one, batch2=C, a_bs2=(T + KW) * dg, out_bs2=T * D, tile=tile, **kw)
    per = torch.zeros(C * T, D, device=DEV)
    for c in range(C):
        ops.gemm(xg[c * (T + KW):], wt, T, dg, KW * dg, resid=resid[c * T:], out_f32=per[c * T:], **kw)
    torch.cuda.synchronize()
    assert torch.equal(one, per)
    # reference: conv as written (weights back to [out, in, tap] per group; operands as the kernel sees them)
    xr = xg.float().reshape(G, C, T + KW, dg) if ops.is_split(dtype) else xg.float().reshape(G, C, T + KW, dg)
    wr = wt.float().reshape(G, dg, KW, dg)
    ref = torch.zeros(C, T, D, dtype=torch.float64)
    for gi in range(G):
        win = xr[gi].double().cpu().unfold(1, KW, 1)[:, :T]                     # [C, T, dg_in, KW]
        ref[:, :, gi * dg:(gi + 1) * dg] = torch.einsum("ctik,oki->cto", win, wr[gi].double().cpu())
    ref = F.gelu(ref + bias.double().cpu()) + resid.double().cpu().reshape(C, T, D)
    assert rel(one.reshape(C, T, D), ref) < (2e-5 if dtype != BF16 else 2e-5)


def test_gemm_split_k_argument_checks():
    from fdm_amd._lib import FdmError
    A = torch.zeros(64, 1024, device=DEV, dtype=torch.bfloat16); Wt = torch.zeros(128, 1024, device=DEV, dtype=torch.bfloat16)
    out = torch.zeros(4, 64, 128, device=DEV)
    for kw in (dict(ksplit=3), dict(ksplit=5), dict(ksplit=2, act=ACT_RELU), dict(ksplit=2, tile=3), dict(ksplit=2, out_t=torch.zeros(64, 128, device=DEV, dtype=torch.bfloat16)),
               dict(ksplit=2, ksplit_stride=100)):
        kw.setdefault("ksplit_stride", 64 * 128)
        with pytest.raises(FdmError):
            ops.gemm(A, Wt, 64, 128, 1024, out_f32=out, **kw)
    with pytest.raises(FdmError):
        ops.layernorm(out, torch.ones(128, device=DEV), torch.zeros(128, device=DEV), 64, 256, y_f32=out, x_planes=2, x_plane_stride=8)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("M,N,K,act", [(7, 256, 256, ACT_MISH), (100, 1024, 1024, ACT_NONE), (800, 3072, 1024, ACT_RELU),
                                       (33, 15069, 1024, ACT_NONE), (257, 192, 2048, ACT_GELU_ERF),
                                       (1600, 2048, 1024, ACT_GELU_TANH), (130, 1024, 5120, ACT_LEAKY02)])
def test_gemm(dtype, M, N, K, act):
    g = torch.Generator().manual_seed(M * 7 + N)
    td = ops.tdtype(dtype)
    A = torch.randn(M, K, generator=g).to(td)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(td)
    bias = torch.randn(N + 3, generator=g)[:N].contiguous()
    resid = torch.randn(M, N, generator=g)
    ref = act_ref(A.float() @ W.float().t() + bias, act) + resid
    o32 = torch.zeros(M, N, device=DEV)
    ot = torch.zeros(M, N, device=DEV, dtype=td)
    ops.gemm(A.to(DEV), W.to(DEV), M, N, K, bias=bias.to(DEV), act=act, resid=resid.to(DEV), out_f32=o32, out_t=ot)
    torch.cuda.synchronize()
    tol = 2e-5 if dtype == F32 else 1e-2
    assert rel(o32, ref) < tol
    assert rel(ot.float(), ref) < (tol if dtype == F32 else 2e-2)


@pytest.mark.parametrize("dtype,tol", [(F16X3, 2e-6)])
@pytest.mark.parametrize("M,N,K,act,tile", [(7, 256, 256, ACT_MISH, 0), (100, 1024, 1024, ACT_NONE, 0), (800, 3072, 1024, ACT_RELU, 0),
                                            (33, 1500, 1024, ACT_NONE, 6), (257, 192, 2048, ACT_GELU_ERF, 9),
                                            (1600, 2048, 1024, ACT_NONE, 3), (530, 1024, 2048, ACT_LEAKY02, 2), (800, 1024, 1024, ACT_NONE, 8)])
def test_gemm_split_operands(dtype, tol, M, N, K, act, tile):
    """Split-operand GEMMs (hi/lo planes, three 16-bit MFMA passes) against an fp64 product of the SAME fp32 inputs:
    f16x3 is fp32-class (the fp32 MFMA kernel itself sits at ~1e-6 on this scale)."""
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N + 3, generator=g)[:N].contiguous()
    resid = torch.randn(M, N, generator=g)
    ref = (act_ref(A.double() @ W.double().t() + bias, act) + resid)
    As, Ws = ops.to_operand(A.to(DEV), dtype), ops.to_operand(W.to(DEV), dtype)
    # the plane pairs reproduce the fp32 inputs to 2^-22
    assert rel(As.float(), A) < 1e-6
    o32 = torch.zeros(M, N, device=DEV)
    ot = ops.Split.empty(M, N, dtype, DEV)
    ops.gemm(As, Ws, M, N, K, bias=bias.to(DEV), act=act, resid=resid.to(DEV), out_f32=o32, out_t=ot, tile=tile)
    torch.cuda.synchronize()
    assert rel(o32, ref) < tol
    assert rel(ot.float(), o32) < 1e-6      # the output plane pair round-trips the fp32 result
    # tile choice changes speed only: identical bits from another tile
    o2 = torch.zeros(M, N, device=DEV)
    ops.gemm(As, Ws, M, N, K, bias=bias.to(DEV), act=act, resid=resid.to(DEV), out_f32=o2, tile=1)
    assert torch.equal(o2, o32)


@pytest.mark.parametrize("dtype", [BF16, F32])
def test_pingpong_tile_is_bit_identical_and_race_free(dtype):
    """FDM_TILE_256x128_PP (two wave groups half a period apart, counted vmcnt across raw barriers): the same bits as the 64x64
    tile on interior and edge shapes -- one k-tile (prologue == tail), two, odd row counts, N not a tile multiple (general
    epilogue), K up to 4096 -- and over 10 repeated launches each (a landing or reuse race shows as a run-to-run difference)."""
    from fdm_amd._lib import TILE_256x128_PP, TILE_64x64
    g = torch.Generator().manual_seed(77)
    k_unit = 64 if dtype == BF16 else 32
    shapes = [(256, 128, k_unit), (300, 256, 2 * k_unit), (515, 384, 3 * k_unit), (1000, 200, 5 * k_unit), (2049, 1024, 1024), (6400, 1024, 2048),
              (777, 3072, 1024), (4096, 512, 4096)]
    for (M, N, K) in shapes:
        A = ops.to_operand(torch.randn(M, K, generator=g).to(DEV), dtype)
        W = ops.to_operand((torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV), dtype)
        bias = torch.randn(N, generator=g).to(DEV)
        res = torch.randn(M, N, generator=g).to(DEV)
        ref = torch.zeros(M, N, device=DEV)
        ops.gemm(A, W, M, N, K, bias=bias, act=ACT_RELU, resid=res, out_f32=ref, tile=TILE_64x64)
        for rep in range(10):
            out = torch.full((M, N), float("nan"), device=DEV)
            ops.gemm(A, W, M, N, K, bias=bias, act=ACT_RELU, resid=res, out_f32=out, tile=TILE_256x128_PP)
            assert torch.equal(out, ref), f"dtype {dtype} shape {(M, N, K)} rep {rep}: max diff {float((out - ref).abs().max())}"


def test_pingpong_tile_batched_launch_and_kv_epilogue():
    """The ping-pong tile through the other launch forms the step uses: a batched GEMM (blockIdx.z: the CFG latent encoder,
    shared A, per-half addend and output) and a QKV projection with the packed K / V epilogue -- the same bits as the 64x64 tile."""
    from fdm_amd._lib import TILE_256x128_PP, TILE_64x64
    g = torch.Generator().manual_seed(8)
    M, N, K = 600, 512, 512
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(2 * M, N, generator=g).to(DEV)
    outs = []
    for tile in (TILE_64x64, TILE_256x128_PP):
        o32 = torch.zeros(2 * M, N, device=DEV)
        ot = torch.zeros(2 * M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm(A, W, M, N, K, bias=bias, act=ACT_MISH, resid=res, out_f32=o32, out_t=ot, batch=2, out_bs=M * N, tile=tile)
        outs.append((o32, ot))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert rel(outs[1][0][M:], act_ref(A.float().cpu().double() @ W.float().cpu().double().t() + bias.cpu(), ACT_MISH) + res[M:].cpu()) < 2e-2
    B, H, L, hd = 3, 4, 200, 128
    d = H * hd
    x = torch.randn(B * L, d, generator=g).to(torch.bfloat16).to(DEV)
    Wqkv = (torch.randn(3 * d, d, generator=g) / math.sqrt(d)).to(torch.bfloat16).to(DEV)
    bq = (0.1 * torch.randn(3 * d, generator=g)).to(DEV)
    packs = []
    for tile in (TILE_64x64, TILE_256x128_PP):
        q_t = torch.zeros(B * L, d, device=DEV, dtype=torch.bfloat16)
        kp, vp, Lpad = ops.kv_buffers(B, H, L, hd, torch.bfloat16, DEV)
        ops.gemm(x, Wqkv, B * L, 3 * d, d, bias=bq, out_t=q_t, ldo_t=d, out_kp=kp, kp_col0=d, out_vp=vp, vp_col0=2 * d, kv_L=L, kv_Lpad=Lpad,
                 kv_hd=hd, tile=tile)
        packs.append((q_t, kp, vp))
    for a_, b_ in zip(packs[0], packs[1]):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("dtype", [BF16, F32, F16X3])
def test_one_round_tiles_are_bit_identical(dtype):
    """FDM_TILE_80x128 (ten A pieces over eight waves: the first two waves carry one more LDS-DMA piece per stage and wait on
    their own count) and FDM_TILE_64x128, sized so that 800 rows fill the 256 CUs in ONE round (240 / 208 workgroups): the
    same bits as the 64x64 tile on interior and ragged shapes, repeated (a wait-count slip shows as a run-to-run difference),
    and through the QKV projection's packed K / V epilogue where an 80-row tile straddles the 200-frame clip boundary."""
    from fdm_amd._lib import TILE_80x128, TILE_64x128, TILE_64x64
    g = torch.Generator().manual_seed(80)
    k_unit = 32 if dtype == F32 else 64
    shapes = [(80, 128, k_unit), (800, 3072, 1024), (800, 2048, 1024), (801, 200, 3 * k_unit), (79, 384, 2 * k_unit), (1992, 1024, 2048), (2400, 3072, 1024)]
    for (M, N, K) in shapes:
        A = ops.to_operand(torch.randn(M, K, generator=g).to(DEV), dtype)
        W = ops.to_operand((torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV), dtype)
        bias = torch.randn(N, generator=g).to(DEV)
        res = torch.randn(M, N, generator=g).to(DEV)
        ref = torch.zeros(M, N, device=DEV)
        ops.gemm(A, W, M, N, K, bias=bias, act=ACT_GELU_ERF, resid=res, out_f32=ref, tile=TILE_64x64)
        for tile in (TILE_80x128, TILE_64x128):
            for rep in range(5):
                out = torch.full((M, N), float("nan"), device=DEV)
                ops.gemm(A, W, M, N, K, bias=bias, act=ACT_GELU_ERF, resid=res, out_f32=out, tile=tile)
                assert torch.equal(out, ref), f"dtype {dtype} tile {tile} shape {(M, N, K)} rep {rep}"
    if dtype == F32:
        return
    B, H, L, hd = 4, 4, 200, 128
    d = H * hd
    x32 = torch.randn(B * L, d, generator=g).to(DEV)
    w32 = (torch.randn(3 * d, d, generator=g) / math.sqrt(d)).to(DEV)
    x, Wqkv = ops.to_operand(x32, dtype), ops.to_operand(w32, dtype)
    bq = (0.1 * torch.randn(3 * d, generator=g)).to(DEV)
    packs = []
    for tile in (TILE_64x64, TILE_80x128, TILE_64x128):
        q_t = ops.Split.empty(B * L, d, dtype, DEV) if dtype == F16X3 else torch.zeros(B * L, d, device=DEV, dtype=torch.bfloat16)
        if dtype == F16X3:
            Lpad = ops.kv_pad(L)
            kp = ops.Split(torch.zeros(2, B * H, Lpad * hd, device=DEV, dtype=torch.float16), F16X3)
            vp = ops.Split(torch.zeros(2, B * H, Lpad * hd, device=DEV, dtype=torch.float16), F16X3)
        else:
            kp, vp, Lpad = ops.kv_buffers(B, H, L, hd, torch.bfloat16, DEV)
        ops.gemm(x, Wqkv, B * L, 3 * d, d, bias=bq, out_t=q_t, ldo_t=d, out_kp=kp, kp_col0=d, out_vp=vp, vp_col0=2 * d, kv_L=L, kv_Lpad=Lpad,
                 kv_hd=hd, tile=tile)
        packs.append(tuple(t.planes if dtype == F16X3 else t for t in (q_t, kp, vp)))
    for other in packs[1:]:
        for a_, b_ in zip(packs[0], other):
            assert torch.equal(a_, b_)


@pytest.mark.parametrize("dtype", [BF16, F32, F16X3])
def test_every_tile_on_random_ragged_shapes(dtype):
    """All twelve FDM_TILE_* ids on 20 seeded random shapes (rows and columns that are not tile multiples, K from one k-tile up,
    random activation, with / without residual and bias): the general (edge-handling) epilogue of every tile gives the bits of
    the 64x64 tile, and those are within the kind's tolerance of an fp64 product."""
    import random
    rnd = random.Random(1234 + dtype)
    g = torch.Generator().manual_seed(99)
    k_unit = 32 if dtype == F32 else 64
    acts = [ACT_NONE, ACT_RELU, ACT_GELU_ERF, ACT_MISH, ACT_LEAKY02]
    for case in range(20):
        M, N, K = rnd.randint(1, 700), rnd.choice([rnd.randint(1, 90) * 4, rnd.randint(1, 12) * 128]), k_unit * rnd.randint(1, 9)
        act, use_res, use_bias = rnd.choice(acts), rnd.random() < 0.5, rnd.random() < 0.7
        A32 = torch.randn(M, K, generator=g)
        W32 = torch.randn(N, K, generator=g) / math.sqrt(K)
        A, W = ops.to_operand(A32.to(DEV), dtype), ops.to_operand(W32.to(DEV), dtype)
        bias = torch.randn(N, generator=g).to(DEV) if use_bias else None
        res = torch.randn(M, N, generator=g).to(DEV) if use_res else None
        ref = torch.zeros(M, N, device=DEV)
        ops.gemm(A, W, M, N, K, bias=bias, act=act, resid=res, out_f32=ref, tile=1)
        want = act_ref(A32.double() @ W32.double().t() + (bias.cpu().double() if use_bias else 0.0), act) + (res.cpu().double() if use_res else 0.0)
        assert rel(ref, want) < (2e-2 if dtype == BF16 else 2e-5), f"case {case}: {(M, N, K)} act {act}"
        for tile in range(2, 13):
            out = torch.full((M, N), float("nan"), device=DEV)
            ops.gemm(A, W, M, N, K, bias=bias, act=act, resid=res, out_f32=out, tile=tile)
            assert torch.equal(out, ref), f"case {case}: tile {tile} shape {(M, N, K)} act {act} res {use_res} bias {use_bias}"


def test_split_producers_write_plane_pairs():
    """LayerNorm, scheduler and fp32 attention write GEMM inputs as plane pairs in the split modes."""
    g = torch.Generator().manual_seed(3)
    M, d = 37, 1024
    x = torch.randn(M, d, generator=g)
    gam, bet = torch.randn(d, generator=g), torch.randn(d, generator=g)
    y32 = torch.zeros(M, d, device=DEV)
    yt = ops.Split.empty(M + 5, d, F16X3, DEV)
    ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV), M, d, y_f32=y32, y_t=yt[5:], dtype=F16X3)
    assert rel(yt.float()[5:], y32) < 1e-6 and rel(y32, F.layer_norm(x, (d,), gam, bet)) < 1e-5
    n = M * d
    x0, xx = torch.randn(n, generator=g), torch.randn(n, generator=g)
    c = torch.rand(4, generator=g)
    xo = torch.zeros(n, device=DEV)
    xot = ops.Split.empty(M, d, F16X3, DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    tseq = torch.tensor([2], dtype=torch.int32, device=DEV)
    ops.sched_step(0, x0.to(DEV), xx.to(DEV), xo, n, tseq=tseq, step=step, c1=c.to(DEV), c2=c.to(DEV), sigma=c.to(DEV),
                   noise=torch.zeros(n, device=DEV), x_out_t=xot)
    assert rel(xot.float().reshape(-1), xo) < 1e-6


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_gemm_overlapping_rows_is_strided_conv(dtype):
    """lda < K: Conv1d(512, 512, k=3, stride=2) over a channels-last signal, no im2col."""
    g = torch.Generator().manual_seed(5)
    td = ops.tdtype(dtype)
    T, Cc, k, s = 101, 512, 3, 2
    x = torch.randn(T, Cc, generator=g).to(td)
    w = (torch.randn(Cc, Cc, k, generator=g) / math.sqrt(Cc * k)).to(td)
    ref = F.conv1d(x.float().t().unsqueeze(0), w.float(), stride=s)[0].t()
    To = (T - k) // s + 1
    wr = w.permute(0, 2, 1).contiguous().view(Cc, k * Cc)
    out = torch.zeros(To, Cc, device=DEV)
    ops.gemm(x.to(DEV), wr.to(DEV), To, Cc, k * Cc, lda=s * Cc, out_f32=out)
    torch.cuda.synchronize()
    assert rel(out, ref) < (2e-5 if dtype == F32 else 1e-2)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_gemm_batched_column_groups_and_row_broadcast_residual(dtype):
    g = torch.Generator().manual_seed(6)
    td = ops.tdtype(dtype)
    G, M, K, Ng = 4, 50, 128, 64
    A = torch.randn(G, M, K, generator=g).to(td)
    W = (torch.randn(G, Ng, K, generator=g) / math.sqrt(K)).to(td)
    bias = torch.randn(G * Ng, generator=g)
    rrow = torch.randn(1, G * Ng, generator=g)
    ref = torch.cat([A[i].float() @ W[i].float().t() for i in range(G)], 1) + bias + rrow
    out = torch.zeros(M, G * Ng, device=DEV)
    ops.gemm(A.to(DEV), W.to(DEV), M, Ng, K, batch=G, a_bs=M * K, w_bs=Ng * K, bias=bias.to(DEV), bias_bs=Ng,
             resid=rrow.to(DEV), ldr=G * Ng, resid_row_mod=1, out_f32=out, ldo_f32=G * Ng, out_bs=Ng)
    torch.cuda.synchronize()
    assert rel(out, ref) < (2e-5 if dtype == F32 else 1e-2)


def mha_ref(q, k, v, scale, bias):
    s = torch.einsum("bhid,bhjd->bhij", q, k) * scale
    if bias is not None:
        s = s + bias
    return torch.einsum("bhij,bhjd->bhid", torch.softmax(s, -1), v)


def alibi(H, L, period, slopes):
    i = torch.arange(L).view(L, 1)
    j = torch.arange(L).view(1, L)
    m = -slopes.view(H, 1, 1) * torch.div(i - j, period, rounding_mode="floor").float()
    return m.masked_fill((j > i).unsqueeze(0), float("-inf"))


def pack_ref(k, v, Lpad, dtype):
    """Host restatement of the fragment-packed K / V layouts (include/fdm_hip.h, fdm_attn_args).
    k, v: [B, H, L, hd] -> two [B*H, Lpad*hd] tensors (pad keys zero)."""
    B, H, L, hd = k.shape
    epc = 8 if dtype in (BF16, F16) else 4
    kt_keys = 4 * epc
    nsub, nks = kt_keys // 16, hd // (4 * epc)
    l = torch.arange(L).view(L, 1)
    e = torch.arange(hd).view(1, hd)
    kt, w = l // kt_keys, l % kt_keys
    if nsub == 2:
        sub, r = (w >> 2) & 1, ((w >> 3) << 2) | (w & 3)
    else:
        sub, r = torch.zeros_like(w), w
    ch = e // epc
    koff = (((((kt * nsub + sub) * nks + (ch >> 2)) * 4 + (ch & 3)) * 16 + r) * epc + e % epc).reshape(-1)
    voff = ((((kt * (hd // 16) + (e >> 4)) * 4 + w // epc) * 16 + (e & 15)) * epc + w % epc).reshape(-1)
    assert koff.unique().numel() == L * hd and voff.unique().numel() == L * hd
    kp = torch.zeros(B * H, Lpad * hd, dtype=k.dtype)
    vp = torch.zeros(B * H, Lpad * hd, dtype=v.dtype)
    kp[:, koff] = k.reshape(B * H, L * hd)
    vp[:, voff] = v.reshape(B * H, L * hd)
    return kp, vp


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("B,H,L,hd,causal,period", [(1, 2, 7, 128, True, 30), (2, 8, 100, 128, True, 30), (4, 8, 200, 128, True, 30),
                                                    (1, 4, 600, 128, True, 25), (2, 16, 98, 64, False, 1), (1, 8, 33, 128, False, 1),
                                                    (1, 16, 498, 64, False, 1), (2, 4, 75, 256, True, 25), (2, 4, 130, 128, True, 3),
                                                    (3, 4, 384, 128, True, 30), (1, 8, 500, 64, True, 30)])
def test_attention_via_qkv_gemm_layout(dtype, B, H, L, hd, causal, period):
    """QKV produced by the GEMM (K and V written fragment-packed by its epilogue), then the fused attention kernel."""
    g = torch.Generator().manual_seed(L + hd)
    td = ops.tdtype(dtype)
    d = H * hd
    x = torch.randn(B * L, d, generator=g).to(td)
    Wqkv = (torch.randn(3 * d, d, generator=g) / math.sqrt(d)).to(td)
    bqkv = 0.1 * torch.randn(3 * d, generator=g)
    q_t = torch.zeros(B * L, d, device=DEV, dtype=td)
    kp, vp, Lpad = ops.kv_buffers(B, H, L, hd, td, DEV)
    ops.gemm(x.to(DEV), Wqkv.to(DEV), B * L, 3 * d, d, bias=bqkv.to(DEV), out_t=q_t, ldo_t=d,
             out_kp=kp, kp_col0=d, out_vp=vp, vp_col0=2 * d, kv_L=L, kv_Lpad=Lpad, kv_hd=hd)
    slopes = torch.tensor([2.0 ** (-(i + 1)) for i in range(H)])
    scale = 1.0 / math.sqrt(hd) if causal else (1.0 / 32 if hd == 128 else 0.125)
    o = torch.zeros(B * L, d, device=DEV, dtype=td)
    ops.attention(q_t, kp, vp, o, B=B, H=H, L=L, hd=hd, ldq=d, ldo=d, Lpad=Lpad,
                  scale=scale, causal=causal, slopes=slopes.to(DEV) if causal else None, period=period)
    torch.cuda.synchronize()
    qkv = (x.float() @ Wqkv.float().t() + bqkv)
    if dtype == BF16:
        qkv = qkv.to(td).float()
    q, k, v = [t.view(B, L, H, hd).transpose(1, 2) for t in qkv.split(d, 1)]
    # GEMM-produced packed K / V match the host restatement of the layout
    kref, vref = pack_ref(k, v, Lpad, dtype)
    tol = 2e-5 if dtype == F32 else 1e-2
    assert rel(q_t.float(), qkv[:, :d]) < tol
    assert rel(kp.float(), kref) < tol and rel(vp.float(), vref) < tol
    ref = mha_ref(q, k, v, scale, alibi(H, L, period, slopes) if causal else None).transpose(1, 2).reshape(B * L, d)
    assert rel(o.float(), ref) < (3e-5 if dtype == F32 else 2e-2)


@pytest.mark.parametrize("B,H,L,hd,causal,period", [(2, 8, 200, 128, True, 30), (1, 4, 37, 128, True, 25), (2, 3, 45, 64, False, 1),
                                                    (1, 2, 600, 128, True, 30), (3, 2, 8, 128, True, 30),
                                                    (2, 4, 200, 256, True, 25), (1, 2, 75, 256, True, 25), (1, 1, 600, 256, False, 1)])
def test_split_attention_via_qkv_gemm(B, H, L, hd, causal, period):
    """FDM_F16X3: the QKV GEMM writes Q and the packed K / V as fp16 plane pairs, the split attention kernel runs both
    products in three 16-bit MFMA passes (probabilities split in registers) and writes O as a plane pair.  Checked against an
    fp64 evaluation of the same fp32 inputs at the fp32 kernel's own tolerance.  head_dim 256 (BIWI) holds a key tile's K, then its V,
    at one wave per SIMD (attention.hpp, attn_streamed)."""
    g = torch.Generator().manual_seed(L + hd)
    d = H * hd
    x = torch.randn(B * L, d, generator=g)
    Wqkv = torch.randn(3 * d, d, generator=g) / math.sqrt(d)
    bqkv = 0.1 * torch.randn(3 * d, generator=g)
    xs, ws = ops.to_operand(x.to(DEV), F16X3), ops.to_operand(Wqkv.to(DEV), F16X3)
    Lpad = ops.kv_pad(L)
    q = ops.Split.empty(B * L, d, F16X3, DEV)
    kp = ops.Split(torch.zeros(2, B * H, Lpad * hd, device=DEV, dtype=torch.float16), F16X3)
    vp = ops.Split(torch.zeros(2, B * H, Lpad * hd, device=DEV, dtype=torch.float16), F16X3)
    ops.gemm(xs, ws, B * L, 3 * d, d, bias=bqkv.to(DEV), out_t=q, ldo_t=d, out_kp=kp, kp_col0=d, out_vp=vp, vp_col0=2 * d,
             kv_L=L, kv_Lpad=Lpad, kv_hd=hd)
    slopes = torch.tensor([2.0 ** (-(i + 1)) for i in range(H)])
    scale = 1.0 / math.sqrt(hd) if causal else (0.125 if hd == 64 else 1.0 / math.sqrt(hd))
    o = ops.Split.empty(B * L, d, F16X3, DEV)
    ops.attention(q, kp, vp, o, B=B, H=H, L=L, hd=hd, ldq=d, ldo=d, Lpad=Lpad, scale=scale, causal=causal,
                  slopes=slopes.to(DEV) if causal else None, period=period)
    torch.cuda.synchronize()
    qkv = x.double() @ Wqkv.double().t() + bqkv
    qq, kk, vv = [t.view(B, L, H, hd).transpose(1, 2) for t in qkv.split(d, 1)]
    kref, vref = pack_ref(kk.float(), vv.float(), Lpad, BF16)          # the 16-bit packed layout, per plane
    assert rel(q.float(), qkv[:, :d]) < 2e-6
    assert rel(kp.float(), kref) < 2e-6 and rel(vp.float(), vref) < 2e-6
    ref = mha_ref(qq, kk, vv, scale, alibi(H, L, period, slopes).double() if causal else None).transpose(1, 2).reshape(B * L, d)
    assert rel(o.float(), ref) < 3e-6


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("B,H,L,hd", [(2, 3, 45, 64), (1, 2, 100, 128), (1, 1, 33, 256)])
def test_pack_kv_is_the_documented_permutation(dtype, B, H, L, hd):
    g = torch.Generator().manual_seed(B + L)
    td = ops.tdtype(dtype)
    d = H * hd
    kv = torch.randn(B * L, 2 * d, generator=g).to(td)
    kp, vp, Lpad = ops.kv_buffers(B, H, L, hd, td, DEV)
    kvd = kv.to(DEV)
    ops.pack_kv(kvd, kvd[:, d:], kp, vp, B=B, H=H, L=L, Lpad=Lpad, hd=hd, ldk=2 * d, ldv=2 * d)
    torch.cuda.synchronize()
    k, v = [t.reshape(B, L, H, hd).transpose(1, 2) for t in kv.split(d, 1)]
    kref, vref = pack_ref(k, v, Lpad, dtype)
    assert torch.equal(kp.cpu(), kref) and torch.equal(vp.cpu(), vref)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("M,d,act", [(5, 256, ACT_NONE), (800, 1024, ACT_NONE), (333, 512, ACT_GELU_ERF)])
def test_layernorm_with_addends(dtype, M, d, act):
    g = torch.Generator().manual_seed(M + d)
    x = torch.randn(M, d, generator=g) * 2 + 0.3
    am = torch.randn(M, d, generator=g)
    tab = torch.randn(10, d, generator=g)
    gamma = 1 + 0.1 * torch.randn(d, generator=g)
    beta = 0.1 * torch.randn(d, generator=g)
    tidx = torch.tensor([3, 7, 9], dtype=torch.int32)
    step = torch.tensor([1], dtype=torch.int32)
    ref = act_ref(F.layer_norm(x + (am + tab[7]), (d,), gamma, beta, 1e-5), act)
    y32 = torch.zeros(M, d, device=DEV)
    yt = torch.zeros(M, d, device=DEV, dtype=ops.tdtype(dtype))
    ops.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV), M, d, add_mat=am.to(DEV), add_tab=tab.to(DEV),
                  tab_index=tidx.to(DEV), tab_step=step.to(DEV), act=act, y_f32=y32, y_t=yt, dtype=dtype)
    torch.cuda.synchronize()
    assert rel(y32, ref) < 1e-5
    assert rel(yt.float(), ref) < (1e-5 if dtype == F32 else 1e-2)


def test_sched_ddpm_ddim_cfg_bit_exact_vs_unfused_torch():
    g = torch.Generator().manual_seed(11)
    n = 4 * 1000
    x0, x0u, x, z = [torch.randn(n, generator=g) for _ in range(4)]
    tab = [torch.rand(1000, generator=g) + 0.1 for _ in range(7)]
    tseq = torch.tensor([999, 500, 1, 0], dtype=torch.int32)
    for k in range(4):
        t = int(tseq[k])
        step = torch.tensor([k], dtype=torch.int32, device=DEV)
        out = torch.zeros(n, device=DEV)
        noise = torch.stack([z * (i + 1) for i in range(4)])
        ops.sched_step(0, x0.to(DEV), x.to(DEV), out, n, tseq=tseq.to(DEV), step=step, c1=tab[0].to(DEV),
                       c2=tab[1].to(DEV), sigma=tab[2].to(DEV), noise=noise.to(DEV))
        ref = tab[0][t] * x0 + tab[1][t] * x
        if t > 0:
            ref = ref + tab[2][t] * noise[k]
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), ref), f"ddpm t={t}"
        # DDIM + CFG mix
        out2 = torch.zeros(n, device=DEV)
        ops.sched_step(1, x0.to(DEV), x.to(DEV), out2, n, x0u=x0u.to(DEV), cfg_scale=2.5, tseq=tseq.to(DEV), step=step,
                       sra=tab[3].to(DEV), srm1=tab[4].to(DEV), sqrt_an=tab[5].to(DEV), c_n=tab[6].to(DEV))
        mix = x0u + 2.5 * (x0 - x0u)
        eps = (tab[3][t] * x - mix) / tab[4][t]
        ref2 = mix * tab[5][k] + tab[6][k] * eps
        torch.cuda.synchronize()
        assert torch.equal(out2.cpu(), ref2), f"ddim t={t}"
    # advance increments the device counter
    step = torch.tensor([0], dtype=torch.int32, device=DEV)
    out = torch.zeros(n, device=DEV)
    ops.sched_step(2, x0.to(DEV), None, out, n, x0u=x0u.to(DEV), cfg_scale=2.5, step=step, advance=1)
    torch.cuda.synchronize()
    assert int(step[0]) == 1 and torch.equal(out.cpu(), x0u + 2.5 * (x0 - x0u))


def test_sched_philox_noise_statistics_and_clip_keying():
    n_per_clip, B = 64 * 1024, 4
    n = n_per_clip * B
    zeros = torch.zeros(n, device=DEV)
    one = torch.ones(1000, device=DEV)
    zero = torch.zeros(1000, device=DEV)
    tseq = torch.tensor([5], dtype=torch.int32, device=DEV)
    step = torch.tensor([0], dtype=torch.int32, device=DEV)

    def draw(clip0, Bn, seed=1234):
        out = torch.zeros(n_per_clip * Bn, device=DEV)
        ops.sched_step(0, zeros[: n_per_clip * Bn], zeros[: n_per_clip * Bn], out, n_per_clip * Bn, n_per_clip=n_per_clip,
                       tseq=tseq, step=step, c1=zero, c2=zero, sigma=one, seed=seed, clip0=clip0)
        torch.cuda.synchronize()
        return out.cpu()
    z = draw(0, B)
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01
    assert abs(float((z ** 4).mean()) - 3.0) < 0.1
    # a rank that owns clips [2, 4) draws exactly the noise the single-GPU run gives those clips
    assert torch.equal(draw(2, 2), z[2 * n_per_clip:])
    assert not torch.equal(draw(0, 1, seed=99), z[:n_per_clip])


def test_small_ops():
    g = torch.Generator().manual_seed(2)
    # cast / bias_act / add_rows / small_linear
    a = torch.randn(1000, 256, generator=g)
    v = torch.randn(256, generator=g)
    o = torch.zeros(1000, 256, device=DEV)
    ops.bias_act(a.to(DEV), v.to(DEV), o, 1000, 256, ACT_MISH)
    assert rel(o, F.mish(a + v)) < 1e-6
    B, L, d = 3, 10, 256
    pe, st, em = torch.randn(L, d, generator=g), torch.randn(B, d, generator=g), torch.randn(B, d, generator=g)
    o = torch.zeros(B * L, d, device=DEV)
    ops.add_rows(o, B * L, d, pe.to(DEV), 1, L, st.to(DEV), L, B, em.to(DEV), L, B)
    ref = (pe.unsqueeze(0) + st.unsqueeze(1) + em.unsqueeze(1)).reshape(B * L, d)
    assert rel(o, ref) < 1e-6
    x = torch.eye(8)[[2, 5, 2]]
    Ws, bs = torch.randn(d, 8, generator=g), torch.randn(d, generator=g)
    o = torch.zeros(3, d, device=DEV)
    ops.small_linear(x.to(DEV), Ws.to(DEV), bs.to(DEV), o, 3, 8, d)
    assert rel(o, x @ Ws.t() + bs) < 1e-6
    # replicate pad / group pad
    xin = torch.randn(2, 5, 64, generator=g)
    o = torch.zeros(2, 9, 64, device=DEV)
    ops.pad_rows(xin.to(DEV), o, 2, 5, 64, 2)
    ref = F.pad(xin.transpose(1, 2), (2, 2), mode="replicate").transpose(1, 2)
    assert torch.equal(o.cpu(), ref)
    o = torch.zeros(4, 2, 5 + 6, 16, device=DEV)
    ops.group_pad(xin.to(DEV), o, 2, 5, 64, 4, 3)
    ref = F.pad(xin.view(2, 5, 4, 16).permute(2, 0, 1, 3), (0, 0, 3, 3))
    assert torch.equal(o.cpu(), ref)
    # conv0
    wav = torch.randn(2, 4000, generator=g)
    w0, b0 = torch.randn(512, 1, 10, generator=g), torch.randn(512, generator=g)
    T0 = (4000 - 10) // 5 + 1
    o = torch.zeros(2, T0, 512, device=DEV)
    ops.conv0(wav.to(DEV), w0.to(DEV), b0.to(DEV), o, 2, 4000, T0)
    assert rel(o, F.conv1d(wav.unsqueeze(1), w0, b0, stride=5).transpose(1, 2)) < 1e-5
    # conv0 + LayerNorm(512) + GELU(erf) in one kernel (HuBERT-large's first conv layer), fp32 and bf16 outputs; T0 not a multiple of 16
    gam, bet = torch.randn(512, generator=g), torch.randn(512, generator=g)
    ref = F.gelu(F.layer_norm(F.conv1d(wav.unsqueeze(1), w0, b0, stride=5).transpose(1, 2), (512,), gam, bet, 1e-5))
    o = torch.zeros(2, T0, 512, device=DEV)
    ops.conv0_ln_gelu(wav.to(DEV), w0.to(DEV), b0.to(DEV), gam.to(DEV), bet.to(DEV), o, 2, 4000, T0)
    assert T0 % 16 and rel(o, ref) < 1e-5
    ob = torch.zeros(2, T0, 512, device=DEV, dtype=torch.bfloat16)
    ops.conv0_ln_gelu(wav.to(DEV), w0.to(DEV), b0.to(DEV), gam.to(DEV), bet.to(DEV), ob, 2, 4000, T0)
    assert float((ob.float() - o).abs().max()) <= 2 ** -8 * float(o.abs().max()) + 1e-6      # (the bf16 kind uses the fast GELU: within one bf16 ulp of the fp32 form)
    sp = ops.Split.empty(2 * T0, 512, F16X3, DEV)                                               # split-fp16 plane pair (contract-mode encoder front)
    ops.conv0_ln_gelu(wav.to(DEV), w0.to(DEV), b0.to(DEV), gam.to(DEV), bet.to(DEV), sp, 2, 4000, T0)
    assert float((sp.float().reshape(2, T0, 512) - o).abs().max()) <= 1e-6 * max(1.0, float(o.abs().max()))
    # leaky + instance norm
    xi = torch.randn(2, 37, 1024, generator=g)
    o = torch.zeros(2, 37, 1024, device=DEV)
    ops.leaky_instnorm(xi.to(DEV), 2, 37, 1024, y_f32=o)
    ref = F.instance_norm(F.leaky_relu(xi.transpose(1, 2), 0.2), eps=1e-5).transpose(1, 2)
    assert rel(o, ref) < 1e-5
    # adain
    c, s = torch.randn(12, 11, generator=g), torch.randn(12, 9, generator=g) * 2 + 1
    o = torch.zeros(12, 11, device=DEV)
    ops.adain(c.to(DEV), s.to(DEV), o, 12, 11, 9)
    cm, cs = c.mean(1, keepdim=True), (c.var(1, keepdim=True) + 1e-5).sqrt()
    sm, ss = s.mean(1, keepdim=True), (s.var(1, keepdim=True) + 1e-5).sqrt()
    assert rel(o, (c - cm) / cs * ss + sm) < 1e-5
    torch.cuda.synchronize()


def test_program_record_graph_replay():
    """Record two ops, run eagerly, then replay as a hipGraph with a device-side step counter."""
    n = 4096
    x = torch.zeros(n, device=DEV)
    one = torch.ones(n, device=DEV)
    nz = torch.ones(10, n, device=DEV)          # injected noise is indexed [step k][element]
    tab = torch.arange(1000, dtype=torch.float32, device=DEV)
    zero = torch.zeros(1000, device=DEV)
    onev = torch.ones(1000, device=DEV)
    tseq = torch.arange(10, dtype=torch.int32, device=DEV) * 3
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    prog = ops.Program()
    with prog:
        # x <- tab[t] * 1 + 1 * x, t = tseq[step]; step += 1
        ops.sched_step(0, one, x, x, n, tseq=tseq, step=step, advance=1, c1=tab, c2=onev, sigma=zero, noise=nz)
    assert prog.num_ops == 1
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        prog.run()
        prog.instantiate()
        prog.replay(9)
    s.synchronize()
    assert int(step[0]) == 10
    assert float(x[0]) == float(sum(3 * k for k in range(10)))


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_fused_double_layernorm(dtype):
    g = torch.Generator().manual_seed(77)
    M, d = 301, 1024
    x = torch.randn(M, d, generator=g) * 3
    am = torch.randn(M, d, generator=g)
    tab = torch.randn(4, d, generator=g)
    g1, b1, g2, b2 = [torch.randn(d, generator=g) * 0.1 + (1 if i % 2 == 0 else 0) for i in range(4)]
    ref = F.layer_norm(F.layer_norm(x, (d,), g1, b1, 1e-5) + (am + tab[2]), (d,), g2, b2, 1e-5)
    tidx = torch.tensor([2], dtype=torch.int32, device=DEV)
    y32 = torch.zeros(M, d, device=DEV)
    yt = torch.zeros(M, d, device=DEV, dtype=ops.tdtype(dtype))
    ops.layernorm(x.to(DEV), g1.to(DEV), b1.to(DEV), M, d, add_mat=am.to(DEV), add_tab=tab.to(DEV), tab_index=tidx,
                  y_f32=y32, y_t=yt, dtype=dtype, gamma2=g2.to(DEV), beta2=b2.to(DEV))
    torch.cuda.synchronize()
    assert rel(y32, ref) < 1e-5
    assert rel(yt.float(), ref) < (1e-5 if dtype == F32 else 1e-2)


def test_sched_in_kernel_advance_and_operand_copy():
    n = 256 * 1024 * 4        # many blocks: exercises the last-ticket advance
    x0 = torch.randn(n, device=DEV)
    x = torch.randn(n, device=DEV)
    one = torch.ones(1000, device=DEV)
    zero = torch.zeros(1000, device=DEV)
    tseq = torch.arange(10, dtype=torch.int32, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    arrive = torch.zeros(1, dtype=torch.int32, device=DEV)
    out = torch.zeros(n, device=DEV)
    outt = torch.zeros(n, device=DEV, dtype=torch.bfloat16)
    noise = torch.zeros(5, n, device=DEV)
    for k in range(5):
        ops.sched_step(0, x0, x, out, n, tseq=tseq, step=step, advance=1, c1=one, c2=one, sigma=zero, noise=noise,
                       x_out_t=outt, arrive=arrive)
        torch.cuda.synchronize()
        assert int(step[0]) == k + 1 and int(arrive[0]) == 0
    assert torch.equal(out, x0 + x) and torch.equal(outt, (x0 + x).to(torch.bfloat16))


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("M,N", [(800, 3072), (200, 1024), (1992, 1024), (70, 192)])
def test_gemm_layernorm_folding(dtype, M, N):
    """LayerNorm folded into the GEMMs around it: producer writes per-64-column partial row sums; the consumer
    GEMM takes the RAW rows with W o gamma and corrects in the epilogue; a third GEMM adds LN(raw) as residual."""
    g = torch.Generator().manual_seed(M + N)
    td = ops.tdtype(dtype)
    K = 1024
    tol = 3e-5 if dtype == F32 else 2e-2
    # --- producer: y0 = A0 W0^T + b0 + r0 with statistics of its fp32 output ---
    A0 = torch.randn(M, 256, generator=g).to(td)
    W0 = (torch.randn(K, 256, generator=g) / 16).to(td)
    b0 = torch.randn(K, generator=g) * 0.3
    r0 = torch.randn(M, K, generator=g) + 0.2
    x = torch.zeros(M, K, device=DEV)
    xt = torch.zeros(M, K, device=DEV, dtype=td)
    nparts = K // 64
    stats = torch.full((nparts, M, 2), float("nan"), device=DEV)
    ops.gemm(A0.to(DEV), W0.to(DEV), M, K, 256, bias=b0.to(DEV), resid=r0.to(DEV), out_f32=x, out_t=xt, stat_out=stats)
    torch.cuda.synchronize()
    xr = A0.float() @ W0.float().t() + b0 + r0
    assert rel(x, xr) < tol
    sx = x.cpu().view(M, nparts, 64)
    assert torch.allclose(stats[:, :, 0].cpu().t(), sx.sum(2), rtol=1e-4, atol=1e-3)
    assert torch.allclose(stats[:, :, 1].cpu().t(), (sx ** 2).sum(2), rtol=1e-4, atol=1e-3)
    # --- consumer: LN(x) W^T + b via raw rows ---
    gam = 1 + 0.1 * torch.randn(K, generator=g)
    bet = 0.1 * torch.randn(K, generator=g)
    W = torch.randn(N, K, generator=g) / 32
    b = torch.randn(N, generator=g) * 0.1
    Wp = (W * gam).to(td)
    colsum = Wp.float().sum(1).contiguous()
    bias2 = (W @ bet + b).contiguous()
    ref = F.layer_norm(x.cpu(), (K,), gam, bet, 1e-5) @ W.t() + b
    y = torch.zeros(M, N, device=DEV)
    ops.gemm(xt, Wp.to(DEV), M, N, K, bias=bias2.to(DEV), out_f32=y, ln_stat_in=stats, ln_nparts=nparts, ln_dim=K,
             ln_colsum=colsum.to(DEV))
    torch.cuda.synchronize()
    assert rel(y, ref) < (1e-4 if dtype == F32 else 3e-2)
    # --- residual = LN(raw rows) on the fly ---
    A2 = torch.randn(M, 128, generator=g).to(td)
    W2 = (torch.randn(K, 128, generator=g) / 12).to(td)
    ref2 = A2.float() @ W2.float().t() + F.layer_norm(x.cpu(), (K,), gam, bet, 1e-5)
    y2 = torch.zeros(M, K, device=DEV)
    ops.gemm(A2.to(DEV), W2.to(DEV), M, K, 128, resid=x, out_f32=y2, ln_stat_in=stats, ln_nparts=nparts, ln_dim=K,
             rln_gamma=gam.to(DEV), rln_beta=bet.to(DEV))
    torch.cuda.synchronize()
    assert rel(y2, ref2) < (3e-5 if dtype == F32 else 1e-2)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("mode", [0, 1])
def test_gemm_fused_scheduler_equals_gemm_then_sched_step(dtype, mode):
    """fdm_gemm_args.sched_fuse: the latent-decoder GEMM applying the DDPM (Philox noise) / DDIM update in its epilogue
    writes the same bits as the GEMM followed by fdm_op_sched_step."""
    g = torch.Generator().manual_seed(7 + mode)
    td = ops.tdtype(dtype)
    B, L, d, K = 3, 50, 256, 256
    M = B * L
    A = torch.randn(M, K, generator=g).to(td).to(DEV)
    W = (torch.randn(d, K, generator=g) / math.sqrt(K)).to(td).to(DEV)
    bias = torch.randn(d, generator=g).to(DEV)
    x = torch.randn(M, d, generator=g).to(DEV)
    tab = [(torch.rand(1000, generator=g) + 0.1).to(DEV) for _ in range(7)]
    tseq = torch.tensor([999, 500, 0], dtype=torch.int32, device=DEV)
    for k in range(3):
        step = torch.tensor([k, 0], dtype=torch.int32, device=DEV)
        kw = dict(n_per_clip=L * d, tseq=tseq, step=step, seed=99, clip0=4)
        kw.update(dict(c1=tab[0], c2=tab[1], sigma=tab[2]) if mode == 0 else dict(sra=tab[3], srm1=tab[4], sqrt_an=tab[5], c_n=tab[6]))
        x0 = torch.zeros(M, d, device=DEV)
        ops.gemm(A, W, M, d, K, bias=bias, out_f32=x0)
        ref, ref_t = torch.zeros(M, d, device=DEV), torch.zeros(M, d, device=DEV, dtype=td)
        ops.sched_step(mode, x0, x, ref, M * d, x_out_t=ref_t, **kw)
        xf, xf_t = x.clone(), torch.zeros(M, d, device=DEV, dtype=td)
        ops.gemm(A, W, M, d, K, bias=bias, resid=xf, out_f32=xf, out_t=xf_t, sched=ops.sched_args(mode, None, None, None, M * d, **kw))
        torch.cuda.synchronize()
        assert torch.equal(xf, ref) and torch.equal(xf_t, ref_t), (mode, k)


def test_specialised_and_general_gemm_kernels_agree_bitwise():
    """The GEMM dispatch picks kernels specialised by what a launch can need (lean / packed-KV / LayerNorm-fold epilogues) when
    the host-side predicate says every tile is interior; FDM_TILE_GENERAL or-ed into fdm_gemm_args.tile forces the general
    kernels.  Same bits either way, over plain / heavy / edge shapes, packed K / V with aligned and unaligned clip lengths, the
    fold's producer and consumers, the fused scheduler -- in every operand kind (tools/kernel_variant_hash.py, one fresh process
    per setting)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hashes = []
    for general in ("0", "1", "2", "3"):      # default | FDM_TILE_GENERAL | FDM_TILE_LOCKSTEP (no loader waves) | both
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_variant_hash.py"), general], cwd=root,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        hashes.append([ln for ln in r.stdout.splitlines() if ln.startswith("variant hash")][0])
    assert hashes[0] == hashes[1] == hashes[2] == hashes[3]


def test_time_groupnorm_single_launch_and_chunked_forms():
    """GroupNorm(num_groups = C) over time + GELU (wav2vec2-base's first conv layer): the single-launch form (short clips, or no
    scratch) and the chunked two-launch form (>= 4096 frames with a scratch buffer: fp64 chunk statistics) against torch, fp32 and
    bf16 outputs; T not a multiple of the chunk, C not a multiple of 64."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(11)
    for (B, T, C) in ((2, 300, 512), (2, 9001, 512), (1, 5000, 96)):
        x = torch.randn(B, T, C, generator=g) * 0.7 + 0.3
        gam, bet = torch.randn(C, generator=g), torch.randn(C, generator=g)
        ref = F.gelu(F.group_norm(x.transpose(1, 2), C, gam, bet, 1e-5)).transpose(1, 2)
        scratch = torch.zeros(B * 64 * C * 2, device=DEV, dtype=torch.float64)
        outs = []
        for sc in (None, scratch):
            o = torch.zeros(B, T, C, device=DEV)
            ops.time_groupnorm(x.to(DEV), gam.to(DEV), bet.to(DEV), B, T, C, y_f32=o, act=ACT_GELU_ERF, scratch=sc)
            assert rel(o, ref) < 1e-5, (B, T, C, sc is not None)
            outs.append(o)
        ob = torch.zeros(B, T, C, device=DEV, dtype=torch.bfloat16)
        ops.time_groupnorm(x.to(DEV), gam.to(DEV), bet.to(DEV), B, T, C, y_t=ob, act=ACT_GELU_ERF, dtype=ops.code_of(ob), scratch=scratch)
        assert float((ob.float() - outs[1]).abs().max()) <= 2 ** -8 * float(outs[1].abs().max()) + 1e-6
        # split-fp16 plane pair (the contract-mode encoder front): hi + lo / 2^11 reproduces the fp32 output to ~2^-22, both forms
        for sc in (None, scratch):
            sp = ops.Split.empty(B * T, C, F16X3, DEV)
            ops.time_groupnorm(x.to(DEV), gam.to(DEV), bet.to(DEV), B, T, C, y_t=sp, act=ACT_GELU_ERF, dtype=F16X3, scratch=sc)
            assert float((sp.float().reshape(B, T, C) - outs[1]).abs().max()) <= 1e-6 * max(1.0, float(outs[1].abs().max())), (B, T, C, sc is not None)
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K,act", [(100, 1024, 1024, ACT_NONE), (800, 3072, 1024, ACT_RELU), (257, 192, 2048, ACT_GELU_ERF), (1992, 2048, 1024, ACT_MISH)])
def test_f16_single_plane_gemm_layernorm_cast(M, N, K, act):
    """FDM_F16 (round 6: single-plane fp16 operands, the split kind's hi plane alone) at the operator level: GEMM with every epilogue
    family, the operand copy a LayerNorm writes, and the cast -- against torch fp32 computed from the fp16-rounded inputs (only
    accumulation and the output rounding differ: 2^-11 relative)."""
    g = torch.Generator().manual_seed(M * 7 + N)
    A32 = torch.randn(M, K, generator=g)
    W32 = torch.randn(N, K, generator=g) / math.sqrt(K)
    A, W = ops.to_operand(A32.to(DEV), F16), ops.to_operand(W32.to(DEV), F16)
    assert A.dtype == torch.float16 and torch.equal(A.cpu(), A32.half())          # the cast is round-to-nearest fp16
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    ref = act_ref(A32.half().float() @ W32.half().float().t() + bias, act) + resid
    o32 = torch.zeros(M, N, device=DEV)
    ot = torch.zeros(M, N, device=DEV, dtype=torch.float16)
    ops.gemm(A, W, M, N, K, bias=bias.to(DEV), act=act, resid=resid.to(DEV), out_f32=o32, out_t=ot)
    assert rel(o32, ref) < 2e-5 + (2e-4 if act in (ACT_GELU_ERF, ACT_MISH) else 0)      # (the 16-bit kinds take the hardware exp forms)
    assert rel(ot.float(), ref) < 1e-3
    # every tile gives the same bits in this kind too
    o2 = torch.zeros(M, N, device=DEV)
    for tile in (1, 2, 3, 12):
        ops.gemm(A, W, M, N, K, bias=bias.to(DEV), act=act, resid=resid.to(DEV), out_f32=o2, tile=tile)
        assert torch.equal(o2, o32), tile
    d = 1024 if N >= 1024 else 256
    x = torch.randn(M, d, generator=g) * 2 + 0.3
    gamma, beta = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    yt = torch.zeros(M, d, device=DEV, dtype=torch.float16)
    y32 = torch.zeros(M, d, device=DEV)
    ops.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV), M, d, y_f32=y32, y_t=yt, dtype=F16)
    assert rel(y32, F.layer_norm(x, (d,), gamma, beta, 1e-5)) < 1e-5 and torch.equal(yt.cpu(), y32.cpu().half())
    big = torch.tensor([1e6, -1e6, 65504.0, 3.0], device=DEV).repeat(4)
    assert torch.equal(ops.to_operand(big, F16).cpu(), torch.tensor([65504.0, -65504.0, 65504.0, 3.0]).repeat(4).half())     # clamped, never inf


@pytest.mark.parametrize("B,H,L,hd,causal,period", [(4, 8, 200, 128, True, 30), (2, 16, 98, 64, False, 1), (2, 4, 75, 256, True, 25)])
def test_f16_single_plane_attention_via_qkv_gemm(B, H, L, hd, causal, period):
    """The FDM_F16 attention kernel behind the QKV GEMM's packed K / V epilogue, against torch on the fp16-rounded Q, K, V."""
    g = torch.Generator().manual_seed(L + hd)
    d = H * hd
    x = torch.randn(B * L, d, generator=g).half()
    Wqkv = (torch.randn(3 * d, d, generator=g) / math.sqrt(d)).half()
    bqkv = 0.1 * torch.randn(3 * d, generator=g)
    q_t = torch.zeros(B * L, d, device=DEV, dtype=torch.float16)
    kp, vp, Lpad = ops.kv_buffers(B, H, L, hd, torch.float16, DEV)
    ops.gemm(x.to(DEV), Wqkv.to(DEV), B * L, 3 * d, d, bias=bqkv.to(DEV), out_t=q_t, ldo_t=d,
             out_kp=kp, kp_col0=d, out_vp=vp, vp_col0=2 * d, kv_L=L, kv_Lpad=Lpad, kv_hd=hd)
    slopes = torch.tensor([2.0 ** (-(i + 1)) for i in range(H)])
    scale = 1.0 / math.sqrt(hd)
    o = torch.zeros(B * L, d, device=DEV, dtype=torch.float16)
    ops.attention(q_t, kp, vp, o, B=B, H=H, L=L, hd=hd, ldq=d, ldo=d, Lpad=Lpad,
                  scale=scale, causal=causal, slopes=slopes.to(DEV) if causal else None, period=period)
    qkv = (x.float() @ Wqkv.float().t() + bqkv).half().float()
    q, k, v = [t.view(B, L, H, hd).transpose(1, 2) for t in qkv.split(d, 1)]
    kref, vref = pack_ref(k, v, Lpad, F16)
    assert rel(q_t.float(), qkv[:, :d]) < 1e-3 and rel(kp.float(), kref) < 1e-3 and rel(vp.float(), vref) < 1e-3
    ref = mha_ref(q, k, v, scale, alibi(H, L, period, slopes) if causal else None).transpose(1, 2).reshape(B * L, d)
    assert rel(o.float(), ref) < 3e-3
