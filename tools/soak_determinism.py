"""Determinism soak: repeated full-length sampling calls must be bit-identical (catches rare races in the LDS-DMA ring,
the recorded-program replay and the tuner's tile switches)."""
import sys, torch
sys.path.insert(0, 'face-diffusion-model_amd'); sys.path.insert(0, '.')
from fdm_amd.denoiser import DenoiserPlan
from fdm_amd._lib import BF16, F16, F16X3, F32
from fdm_amd import synth as W
DEV = 'cuda:0'
# (round 3: + 32 clips per GPU = 6400 rows, where the tuner picks the ping-pong tile; + 8 style conditions per clip; + BIWI in f16x3:
#  the streamed split attention kernel at head_dim 256)
for preset, B, L, T, cfg, dt, reps in (("vocaset", 32, 200, 120, False, BF16, 4), ("vocaset", 1, 100, 99, False, BF16, 4), ("vocaset", 1, 100, 400, "single", BF16, 4), ("mead", 1, 249, 400, "single", F16X3, 3), ("biwi", 4, 200, 250, False, F16X3, 3),
                                       ("vocaset", 4, 200, 1000, False, BF16, 6), ("vocaset", 4, 498, 400, False, BF16, 4),
                                       ("mead", 4, 300, 400, True, BF16, 4), ("vocaset", 4, 200, 300, False, F32, 3),
                                       ("biwi", 4, 200, 250, False, BF16, 4), ("vocaset", 4, 200, 1000, False, F16X3, 4),
                                       ("mead", 4, 300, 400, True, F16X3, 3), ("vocaset", 4, 498, 300, False, F16X3, 3),
                                       ("vocaset", 4, 200, 1000, False, F16, 4), ("mead", 4, 300, 400, True, F16, 3), ("biwi", 4, 200, 250, False, F16, 3)):      # (round 6: the single-plane fp16 kind)
    inp = W.synth_inputs(preset, B, L, seed=2)
    plan = DenoiserPlan(preset, W.make_fdm_weights(preset), dt, DEV)
    if cfg == "single":        # (round 5: the single-clip setting -- K slices of the out-proj / FFN2 GEMMs; every 16-bit GEMM runs the loader-wave loop)
        plan.set("ksplit.out", 2); plan.set("ksplit.ffn2", 4); cfg = False
    hub = inp["hub"][:, :, :768].contiguous() if preset == "biwi" else inp["hub"]
    S = 8 if (B == 1 and L == 100 and T == 99) else 1
    style = torch.eye(8)[:S].repeat(B, 1) if S > 1 else inp["style"]
    plan.prepare(hub, style, inp.get("emo"), L=L, cfg=cfg, n_conds=S)
    plan.tune()
    x = inp["x"].to(DEV).repeat_interleave(S, dim=0)
    ts = list(range(999, 999 - T, -1))
    ref = None
    for r in range(reps):
        out = plan.sample_ddim(x, T) if preset == "biwi" else plan.sample_ddpm(x, ts, seed=11)
        assert torch.isfinite(out).all()
        if ref is None: ref = out.clone()
        assert torch.equal(out, ref), f"{preset} L={L} run {r} differs: max {float((out - ref).abs().max())}"
    print(f"{preset} B={B} S={S} L={L} T={T} cfg={cfg} { {BF16: 'bf16', F32: 'fp32', F16X3: 'f16x3', F16: 'f16'}[dt] }: {reps} runs bit-identical, tiles {plan.tiles}")

# GEMM level: the one-round tiles (uneven LDS-DMA piece split, per-wave wait counts) against the 64x64 tile's bits, many launches
import math
from fdm_amd import ops
from fdm_amd._lib import TILE_64x64, TILE_80x128, TILE_64x128
g = torch.Generator().manual_seed(5)
for dt, name in ((BF16, 'bf16'), (F16, 'f16'), (F16X3, 'f16x3'), (F32, 'fp32')):
    for (M, N, K) in ((800, 3072, 1024), (800, 2048, 1024), (2400, 1536, 512), (815, 3072, 192)):
        A = ops.to_operand(torch.randn(M, K, generator=g).to(DEV), dt)
        Wt = ops.to_operand((torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV), dt)
        ref = torch.zeros(M, N, device=DEV)
        ops.gemm(A, Wt, M, N, K, out_f32=ref, tile=TILE_64x64)
        out = torch.empty(M, N, device=DEV)
        bad = 0
        for tile in (TILE_80x128, TILE_64x128):
            for rep in range(1500):
                ops.gemm(A, Wt, M, N, K, out_f32=out, tile=tile)
                if rep % 50 == 49 or rep < 3:
                    bad += int(not torch.equal(out, ref))
        assert bad == 0, f"{name} {(M, N, K)}: {bad} differing launches"
    print(f"gemm one-round tiles {name}: 4 shapes x 2 tiles x 1500 launches, sampled compares all bit-identical")
