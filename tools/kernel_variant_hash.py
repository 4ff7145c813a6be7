"""One hash over the outputs of a fixed set of GEMM launches that exercise every specialised kernel variant (plain, heavy
activation, packed K / V with aligned and unaligned clip lengths, LayerNorm-fold producer / consumer, fused scheduler, batched).
Run with argument 1 (FDM_TILE_GENERAL or-ed into every tile: general kernels everywhere), 2 (FDM_TILE_LOCKSTEP: the lockstep k loop instead of
the loader-wave form), 3 (both) and 0 (the default dispatch): the hashes must be equal
(tests/test_ops_gpu.py::test_specialised_and_general_gemm_kernels_agree_bitwise)."""
import hashlib
import math
import sys
sys.path.insert(0, 'face-diffusion-model_amd')
import torch
from fdm_amd import ops
from fdm_amd._lib import ACT_MISH, ACT_NONE, ACT_RELU, BF16, F16X3, F32
DEV = 'cuda:0'
h = hashlib.sha256()


_n = [0]


def add(*ts):
    torch.cuda.synchronize()
    for t in ts:
        t = t.planes if isinstance(t, ops.Split) else t
        h.update(t.detach().float().cpu().numpy().tobytes())
    _n[0] += 1
    print("item", _n[0], h.hexdigest()[:12])      # (running digest: the first differing line localises a disagreement)


def opnd(x, dt):
    return ops.to_operand(x.to(DEV), dt)


def out_t(M, N, dt):
    return ops.Split.empty(M, N, dt, DEV) if ops.is_split(dt) else torch.zeros(M, N, device=DEV, dtype=ops.tdtype(dt))


g = torch.Generator().manual_seed(5)
GEN = {'1': 0x100, '2': 0x200, '3': 0x300}.get(sys.argv[1] if len(sys.argv) > 1 else '0', 0)      # include/fdm_hip.h FDM_TILE_GENERAL (0x100), FDM_TILE_LOCKSTEP (0x200)
for dt in (BF16, F32, F16X3):
    for tile in (0 | GEN, 2 | GEN, 3 | GEN, 8 | GEN, 10 | GEN, 9 | GEN, 11 | GEN, 12 | GEN):      # (10 = the ping-pong tile: its launcher honours FDM_TILE_GENERAL too)
        # plain / heavy, interior and edge shapes
        for (M, N, K, act) in ((800, 1024, 1024, ACT_NONE), (130, 2048, 512, ACT_RELU), (64, 1024, 1024, ACT_MISH), (77, 192, 256, ACT_RELU)):
            A, W = opnd(torch.randn(M, K, generator=g), dt), opnd(torch.randn(N, K, generator=g) / math.sqrt(K), dt)
            bias, res = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
            o32, ot = torch.zeros(M, N, device=DEV), (out_t(M, N, dt) if dt != F32 else None)
            ops.gemm(A, W, M, N, K, bias=bias, act=act, resid=res, out_f32=o32, out_t=ot, tile=tile)
            add(o32, *([ot] if ot is not None else []))
        # QKV projection into the packed K / V layouts: clip lengths that are / are not whole packed chunks
        if (dt != F16X3 or (tile & 0xff) in (0, 3, 8)) and (tile & 0xff) not in (9, 11, 12):
            for (B, L) in ((4, 200), (2, 498), (3, 33)):
                d, H = 512, 4
                hd, M = d // H, B * L
                A, W = opnd(torch.randn(M, d, generator=g), dt), opnd(torch.randn(3 * d, d, generator=g) / math.sqrt(d), dt)
                q = out_t(M, d, dt) if dt != F32 else torch.zeros(M, d, device=DEV)
                kvd = torch.float32 if dt == F32 else ops.tdtype(dt)
                Lpad = ops.kv_pad(L)
                if ops.is_split(dt):
                    kp, vp = ops.Split.empty(B * H, Lpad * hd, dt, DEV), ops.Split.empty(B * H, Lpad * hd, dt, DEV)
                else:
                    kp, vp, _ = ops.kv_buffers(B, H, L, hd, kvd, DEV)
                kw = dict(out_f32=q) if dt == F32 else dict(out_t=q)
                ops.gemm(A, W, M, 3 * d, d, ldo_f32=d, ldo_t=d, out_kp=kp, kp_col0=d, out_vp=vp, vp_col0=2 * d, kv_L=L, kv_Lpad=Lpad, kv_hd=hd, tile=tile, **kw)
                add(q, kp, vp)
        # LayerNorm fold: producer statistics, then a consumer
        M, N, K = 400, 1024, 1024
        A, W = opnd(torch.randn(M, K, generator=g), dt), opnd(torch.randn(N, K, generator=g) / math.sqrt(K), dt)
        stats = torch.zeros((N // 64) * M * 2, device=DEV)
        x32, xt = torch.zeros(M, N, device=DEV), (out_t(M, N, dt) if dt != F32 else None)
        ops.gemm(A, W, M, N, K, out_f32=x32, out_t=xt, stat_out=stats, tile=tile)
        W2 = opnd(torch.randn(512, N, generator=g) / math.sqrt(N), dt)
        colsum, gam, bet = torch.randn(512, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
        y = torch.zeros(M, 512, device=DEV)
        ops.gemm(xt if xt is not None else x32, W2, M, 512, N, out_f32=y, ln_stat_in=stats, ln_nparts=N // 64, ln_dim=N, ln_colsum=colsum, tile=tile)
        z = torch.zeros(M, N, device=DEV)
        ops.gemm(A, W, M, N, K, out_f32=z, resid=x32, ln_stat_in=stats, ln_nparts=N // 64, ln_dim=N, rln_gamma=gam, rln_beta=bet, tile=tile)
        add(x32, stats, y, z)
    # fused scheduler update (64x64 tile only), DDPM with injected noise
    M, N, K = 256, 1024, 1024
    A, W = opnd(torch.randn(M, K, generator=g), dt), opnd(torch.randn(N, K, generator=g) / math.sqrt(K), dt)
    x = torch.randn(M, N, generator=g).to(DEV)
    tab = [torch.rand(1000, generator=g).to(DEV) for _ in range(3)]
    noise = torch.randn(M * N, generator=g).to(DEV)
    tseq = torch.tensor([500], dtype=torch.int32, device=DEV)
    step = torch.zeros(2, dtype=torch.int32, device=DEV)
    sc = ops.sched_args(0, None, None, None, M * N, n_per_clip=M * N, tseq=tseq, step=step, c1=tab[0], c2=tab[1], sigma=tab[2], noise=noise)
    ops.gemm(A, W, M, N, K, resid=x, out_f32=x, sched=sc, tile=GEN)
    add(x)
print("variant hash", h.hexdigest())
