"""bench.py -- animated frames/sec of the FDM diffusion-sampling hot path on N MI355X (one node).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2] [--dtype bf16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A bench "step" = one full sampling call over this rank's batch of clips (cfg2: B = 4 clips x L = 200
latent frames, T = 1000 DDPM diffusion steps, in-kernel Philox noise): W untimed calls, then exactly K
timed calls bracketed by barrier + torch.cuda.synchronize(); max over ranks; rank 0 prints ONE JSON line.
value = (N * B * L * K) / elapsed = latent frames fully denoised per wall-second (BASELINE.json metric).
Weak scaling: every rank owns B clips (clip-level batch sharding); the only
collective is the all-gather of the finished latents, inside the timed region.

Default invocation (the line the driver records) carries three more things besides the headline leg (--dtype, default bf16 =
BASELINE.json configs[1]):
  "f16_mode"       the same workload on single-plane fp16 operands (round 6): bf16's speed, ~8x closer to the reference
  "contract_mode"  the same workload timed in f16x3 -- the mode that meets north_star's 1e-4 tolerance on the 16-bit matrix
                   cores -- with its own HIP-event roofline and its own parity figure;
  "parity"         max-abs of the headline mode's plan against tests/golden/chains_vocaset.npz (outputs of the reference itself:
                   DDPM t = 9..0 and 999..990 with injected noise, DDIM 50), with the tolerance that mode states;
  "rccl_ranks"     dist.get_world_size() when launched under torch.distributed.run (1 otherwise).
--config cfg1x8 is the reference sampler's style loop (samples/sample_diffusion_vocaset.py:71-83: 1 clip x 100 frames x 8 style
one-hots, DDIM 100) as ONE condition-batched call, reported beside the sequential loop of eight B = 1 calls.

--config shipped_vocaset | shipped_mead | shipped_biwi is what the reference's OWN callers issue for one test clip, end to end
inside the timed call (samples/sample_diffusion_vocaset.py:71-88: 10 s of audio -> HuBERT-large -> every one of the 8 style
one-hots, DDIM 100 -> quant -> decode; sample_diffusion_mead.py:67-86: HuBERT -> 1000-step DDPM -> EVQ quant -> decode;
sample_diffusion_biwi.py:60-78: wav2vec2-base -> DDIM 50 -> quant -> decode to 23370 vertices), with a per-stage split
("stages_ms") and, for VOCASET, the reference's sequential loop of eight B = 1 pipelines beside the condition-batched call.

roofline_hbm: the HBM-class kernels of the step (SURVEY.md section 8d "report both per kernel class"): the LayerNorm launches and the
scheduler-fused latent decoder's epilogue traffic, algorithmic bytes / profiled duration (the committed rocprofv3 summary of this
configuration and mode) / 8 TB/s.

roofline: the dominant launch is the captured step graph (one diffusion step).  achieved =
algorithmic FLOPs per diffusion step (2*[B*L*(2d^2 + n_layers*(4d^2 + 2*d*FFN)) +
n_layers*B*2*L^2*d]) / average step duration from HIP events recorded on the plan's stream around the
timed region; peak = dense MFMA bf16 (2.5 PFLOP/s) or fp32 (157.3 TFLOP/s).
cpu_baseline: the CPU oracle (kind "port") timed on the host cores of this box on a bounded sample of
the same workload (a few diffusion steps, extrapolated linearly to T), hoisted mode = the same
arithmetic the GPU path executes; the as-written mode (HuBERT-large re-run inside every step, as
models/fdm_vocaset.py:59 does) is reported next to it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "face-diffusion-model_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

CONFIGS = {
    # name: (preset, B per GPU, L, T, sampler, cfg)
    "cfg1": ("vocaset", 1, 100, 50, "ddim", False),
    "cfg2": ("vocaset", 4, 200, 1000, "ddpm", False),
    "cfg3": ("mead", 4, 300, 1000, "ddpm", True),
    "cfg4": ("biwi", 4, 200, 250, "ddim", False),
    "cfg5": ("vocaset", 4, 498, 1000, "ddpm", False),     # end to end: 10 s audio -> HuBERT -> sample -> quant -> decode
    # the reference sampler's own loop: every style one-hot of a clip, same audio (samples/sample_diffusion_vocaset.py:71-83)
    "cfg1x8": ("vocaset", 1, 100, 100, "ddim", False),
}
CONDS = {"cfg1x8": 8, "shipped_vocaset": 8}
# what the reference's samplers issue per test clip: (preset, audio seconds, latent frames, diffusion steps, sampler, audio encoder)
SHIPPED = {
    "shipped_vocaset": ("vocaset", 10.0, 498, 100, "ddim", "hubert"),     # L = HuBERT frames (samples/sample_diffusion_vocaset.py:76)
    "shipped_mead": ("mead", 10.0, 249, 1000, "ddpm", "hubert"),          # L = HuBERT frames // 2 (sample_diffusion_mead.py:79)
    "shipped_biwi": ("biwi", 10.0, 240, 50, "ddim", "wav2vec"),           # L = int(seconds * 24) (sample_diffusion_biwi.py:69)
}
for _k, (_p, _s, _L, _T, _smp, _enc) in SHIPPED.items():
    CONFIGS[_k] = (_p, 1, _L, _T, _smp, False)
# dense TFLOP/s, MI355X_MICROARCH.md.  The split modes run on the 16-bit matrix cores (3 MFMA passes per product) and are
# priced against that peak with the ALGORITHMIC flops (one product per multiply-add), like every other mode.
PEAK = {"bf16": 2500.0, "f32": 157.3, "f16x3": 2500.0, "f16": 2500.0}


def step_flops(p, B, L, cfg):
    d, nl, ffn = p.d, p.n_layers, p.ffn
    f = 2.0 * (B * L * (2 * d * d + nl * (4 * d * d + 2 * d * ffn)) + nl * B * 2 * L * L * d)
    return f * (2 if cfg else 1)


def host_threads(cap=32):
    """Threads the CPU baseline may use: affinity mask and cgroup quota, capped (a 256-thread torch pool on
    a quota-limited container oversubscribes badly)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, cap))


def cpu_baseline(cfg_name, budget_s=24.0):
    """Bounded sample of the same workload on the host cores (rank 0, N = 1 only)."""
    from oracle import fdm_oracle as FO
    from oracle import hubert_oracle as HO
    from oracle import weights as W
    preset, B, L, T, sampler, cfg = CONFIGS[cfg_name]
    cores = host_threads()
    torch.set_num_threads(cores)
    w = W.make_fdm_weights(preset)
    inp = W.synth_inputs(preset, B, L, seed=1)
    emo = inp.get("emo")
    buf = FO.schedule_buffers()
    g = torch.Generator().manual_seed(0)
    x = inp["x"].clone()

    def hoisted_step(xx, t):
        x0 = FO.fdm_forward(w, preset, inp["hub"], t, xx, inp["style"], emo, folded=True)
        if cfg:
            x0 = FO.cfg_mix(x0, FO.fdm_forward(w, preset, inp["hub"], t, xx, inp["style"], torch.zeros_like(emo), folded=True))
        return FO.ddpm_step(buf, x0, xx, t, torch.randn(xx.shape, generator=g))
    # hoisted mode = the arithmetic the GPU path executes; whole diffusion steps until half the budget is spent
    n, t0 = 0, time.perf_counter()
    while n < 1 or (time.perf_counter() - t0 < budget_s * 0.5 and n < T):
        x = hoisted_step(x, T - 1 - n)
        n += 1
    per_step = (time.perf_counter() - t0) / n
    hoisted = B * L / (per_step * T)
    # as-written mode (models/fdm_vocaset.py:59): HuBERT-large inside every denoiser call, full cross-attention,
    # clips looped as B = 1 calls.  ONE clip of ONE step is timed and scaled by B (the loop is per clip).
    t1 = time.perf_counter()
    wh = W.make_hubert_weights(24)
    wav = torch.randn(1, L * W.PRESETS[preset]["pair"] * 320 + 80, generator=g) * 0.1
    t2 = time.perf_counter()
    hub = HO.hubert_forward(wh, wav, 24)
    FO.fdm_forward(w, preset, hub, T - 1, x[:1], inp["style"][:1], None if emo is None else emo[:1], folded=False)
    per_step_aw = (time.perf_counter() - t2) * B * (2 if cfg else 1)
    return {"value": round(hoisted, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} of {T} diffusion steps of the {cfg_name} batch (hoisted+folded fp32 oracle on torch CPU, "
                      f"{per_step * 1e3:.0f} ms/step), extrapolated linearly to {T} steps; as-written mode (HuBERT-large "
                      f"re-run in every step, full cross-attention): 1 clip of 1 step timed x{B} clips = "
                      f"{per_step_aw * 1e3:.0f} ms/step -> {B * L / (per_step_aw * T):.4f} frames/s",
            "as_written_value": round(B * L / (per_step_aw * T), 5)}


def hbm_class_roofline(pm, rel, p, rows, dtype_name):
    """HBM-class entries from a committed profile summary (tools/pmc_report.py): algorithmic bytes per launch / profiled us / 8 TB/s.
    LayerNorm (fdm_amd csrc/elementwise.hpp ln_row_kernel): the fused LN1+LN2 launch reads the fp32 row and the per-clip addend and
    writes the fp32 row + the operand copy, LN3 reads one and writes the two -- (8 + 2 x 4 + 2 x ob) / 2 bytes per element on
    average, ob = bytes of the operand copy (2 for bf16, 4 for a split-fp16 plane pair).  Scheduler update fused into the latent
    decoder's epilogue: 16 B per latent element (3 fp32 reads + 1 write) -- that launch is a GEMM, so its entry
    states the epilogue's share of HBM-class bytes next to the whole launch's duration (an upper bound on the time it can cost)."""
    ob = {"bf16": 2, "f32": 0, "f16x3": 4, "f16": 2}[dtype_name]
    n = rows * p.d
    out = []
    ln = [k for k in pm["kernels"] if "ln_row_kernel" in k["kernel"]]
    if ln:
        k = ln[0]
        b = n * ((4 + 4 + 4 + ob) + (4 + 4 + ob)) / 2.0
        gbs = b / (k["avg_us"] * 1e-6) / 1e9
        out.append({"kernel": "LayerNorm launches (fused LN1+LN2 with the folded cross-attention addend; LN3)", "launches_per_step": k["launches_per_step"],
                    "algorithmic_bytes_per_launch": int(b), "avg_us_profiled": k["avg_us"], "achieved": round(gbs, 1), "frac": round(gbs / 8000.0, 4),
                    "counter_bytes_per_launch": int((k.get("fetch_mb", 0) + k.get("write_mb", 0)) * 1024 * 1024) or None,
                    "traffic_ratio": round((k.get("fetch_mb", 0) + k.get("write_mb", 0)) * 1024 * 1024 / b, 3) or None})      # counter / algorithmic bytes
    dec = [k for k in pm["kernels"] if "gemm" in k["kernel"] and "true, 1>" in k["kernel"].replace(" ", " ") and "false, true" in k["kernel"]]
    if dec:
        k = dec[0]
        b = n * 16.0
        out.append({"kernel": "scheduler update in the latent decoder's GEMM epilogue", "launches_per_step": k["launches_per_step"],
                    "algorithmic_bytes_per_launch": int(b), "avg_us_profiled": k["avg_us"],
                    "achieved": round(b / (k["avg_us"] * 1e-6) / 1e9, 1), "frac": round(b / (k["avg_us"] * 1e-6) / 1e9 / 8000.0, 4),
                    "counter_bytes_per_launch": int((k.get("fetch_mb", 0) + k.get("write_mb", 0)) * 1024 * 1024) or None,
                    "traffic_ratio": None,       # (the launch's counters hold the GEMM's operand traffic too: no epilogue-only figure)
                    "note": "duration of the whole GEMM launch (the plain 64x64 GEMM of the same shape takes "
                            + str(next((q["avg_us"] for q in pm["kernels"] if "gemm" in q["kernel"] and "false, false, 1>" in q["kernel"] and q["grid"] == k["grid"]), None))
                            + " us): the update rides an MFMA-class kernel"})
    return {"bound": "hbm", "peak": 8000.0, "unit": "GB/s", "counters_from": rel, "counters_commit": pm["summary"].get("commit"),
            "counters_date": pm["summary"].get("date"), "kernels": out} if out else None


PARITY_TOL = {"f32": 1e-4, "f16x3": 1e-4, "bf16": 0.15, "f16": 0.02}     # the bars tests/test_denoiser_gpu.py states per mode


def parity_vs_reference(plan, dev):
    """max-abs of THIS plan object against outputs of the reference itself (tests/golden/chains_vocaset.npz, written by
    tests/golden/make_golden.py importing /root/reference): DDPM t = 9..0 and t = 999..990 with the injected noise, after
    every step, and the 50-step DDIM chain.  The caller re-prepares its own shape afterwards."""
    import numpy as np
    from fdm_amd import synth as W
    g = np.load(os.path.join(ROOT, "tests", "golden", "chains_vocaset.npz"))
    L = int(g["L"])
    inp = W.synth_inputs("vocaset", 1, L, seed=7)
    plan.prepare(inp["hub"], inp["style"], None, L=L)
    worst = 0.0
    for name in ("lo", "hi"):
        rec = []
        plan.sample_ddpm(inp["x"].to(dev), g[f"ddpm_{name}_t"].tolist(), noise=torch.from_numpy(g[f"ddpm_{name}_noise"]), record=rec)
        worst = max(worst, float((torch.stack(rec).cpu().double() - torch.from_numpy(g[f"ddpm_{name}_steps"]).double()).abs().max()))
    out = plan.sample_ddim(inp["x"].to(dev), 50).cpu().double()
    worst = max(worst, float((out - torch.from_numpy(g["ddim_50_final"]).double()).abs().max()))
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f16x3", "f16"],
                    help="arithmetic mode: bf16 (throughput; BASELINE configs[1]), f32 (exact fp32 MFMA) and f16x3 (split-fp16 "
                         "operands, three 16-bit MFMA passes) meet the 1e-4 contract")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the contract-mode leg and the parity legs (sweeps, profiles)")
    ap.add_argument("--contract-steps", type=int, default=5, help="timed sampling calls of the contract-mode (f16x3) leg")
    ap.add_argument("--profile-steps", type=int, default=0, help="profiling only: shorten the diffusion chain to this many steps")
    ap.add_argument("--chain-steps", type=int, default=0, help="tests only: shorten the diffusion chain to this many steps, every stage of the call kept "
                                                              "(audio encoder, quant, decode, gather)")
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU instead of the configuration's (row-count sweeps, tests)")
    ap.add_argument("--broadcast-weights", action="store_true",
                    help="N > 1: rank 0's weights are broadcast to every rank (one flattened RCCL broadcast, outside the timed region) "
                         "instead of each rank generating its own copy from the seed")
    ap.add_argument("--plan-set", action="append", default=[], metavar="KEY=VALUE",
                    help="fdm_plan_set on every denoiser plan of the run before its tables are built, e.g. ksplit.out=2, fuse_ln3=1 (A/B measurements)")
    ap.add_argument("--dump", default="", help="tests only: rank 0 saves the gathered output of the last call to this .npy file")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    dist = None
    if world > 1 or os.environ.get("FDM_DIST_FORCE") == "1":      # (FDM_DIST_FORCE: a ONE-rank process group -- the RCCL branch on a 1-GPU box, tests)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("FDM_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; gloo only for 1-GPU dry runs
        kw = {"device_id": torch.device(dev)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    from fdm_amd import presets
    from fdm_amd._lib import DTYPE_NAMES, F32
    from fdm_amd.denoiser import DenoiserPlan
    from fdm_amd.parallel import gather_clips
    from fdm_amd import synth as W

    preset, B, L, T, sampler, cfg = CONFIGS[a.config]
    S = CONDS.get(a.config, 1)              # style conditions per clip in one step program
    if a.profile_steps or a.chain_steps:
        T = a.profile_steps or a.chain_steps
    if a.batch:
        B = a.batch
    p = presets.get(preset)
    weights = W.make_fdm_weights(preset, seed=0 if (rank == 0 or not a.broadcast_weights) else 12345)
    if a.broadcast_weights and dist is not None:      # ranks > 0 start from a DIFFERENT seed: equal results prove the broadcast
        from fdm_amd.parallel import broadcast_state
        broadcast_state(weights, dist, src=0, device=dev if dist.get_backend() == "nccl" else None)
    inp = W.synth_inputs(preset, B * world, L, seed=1)      # global batch; this rank owns clips [rank*B, (rank+1)*B)
    sl = slice(rank * B, (rank + 1) * B)
    emo = inp["emo"][sl] if "emo" in inp else None
    hub = inp["hub"][sl]
    if preset == "biwi":                                    # wav2vec2-base features are 768 wide
        hub = hub[:, :, :768].contiguous()
    style = inp["style"][sl]
    xT = inp["x"][sl].to(dev)
    if S > 1:                                               # every style one-hot of each clip; all conditions start from the clip's x_T
        style = torch.eye(p.n_style)[:S].repeat(B, 1)
        xT = xT.repeat_interleave(S, dim=0)
    # one clip, one condition per call (the reference's bs = 1 callers): the step program's single-clip setting -- K slices of the
    # out-proj / FFN2 GEMMs summed by the LayerNorm launch that follows (fdm_amd/modules.py SINGLE_CLIP_PLAN)
    from fdm_amd.modules import SINGLE_CLIP_PLAN
    # (decided by the CONFIGURATION -- one clip, one condition per call by definition, no --batch override -- never by the row
    #  count a run happens to have: results depend on the split factor, and a clip must not change with its batch or its rank)
    plan_opts = dict(SINGLE_CLIP_PLAN) if (CONFIGS[a.config][1] == 1 and S == 1 and not a.batch) else {}
    for kv in a.plan_set:
        plan_opts[kv.split("=", 1)[0]] = int(kv.split("=", 1)[1])
    shipped = SHIPPED.get(a.config)                         # the reference callers' per-clip workload, end to end
    e2e = a.config == "cfg5" or shipped is not None
    if shipped:
        style = torch.eye(p.n_style)[:S].repeat(B, 1) if S > 1 else torch.eye(p.n_style)[:1].repeat(B, 1)
        emo = torch.eye(p.n_emo)[4:5].repeat(B, 1) if p.n_emo else None      # the demos' default emotion
    ts = list(range(T - 1, -1, -1))
    n_live = T if sampler == "ddpm" else T - 1              # denoiser calls per sampling call (DDIM: the dead pair is skipped)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        """W untimed calls (after ~3 s of load: the same binary reads 2 % lower when the timed region starts within ~1 s of the
        first GPU load), then exactly `steps` timed calls between barrier + synchronize; max over ranks.  Returns (wall seconds,
        HIP-event milliseconds on the launching stream, last output)."""
        t_spin = time.perf_counter()
        while warmup > 0 and time.perf_counter() - t_spin < 3.0:
            fn(False)                  # rank-local only: a time-bounded loop must not contain a collective
            torch.cuda.synchronize()
        for _ in range(warmup):
            fn(True)
        fence()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()          # the library launches on the caller's (current) stream
        out = None
        for _ in range(steps):
            out = fn(True)
        e1.record()
        fence()
        el = time.perf_counter() - t0
        ev_ms = e0.elapsed_time(e1)
        if dist is not None:
            tt = torch.tensor([el], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt[0])
        return el, ev_ms, out

    def run_leg(dtype_name, steps, warmup, want_parity):
        """One arithmetic mode of the configuration: plan, tables, plan-time tuning, the timed calls, parity of the same plan."""
        dt = DTYPE_NAMES[dtype_name]
        plan = DenoiserPlan(preset, weights, dt, dev)
        for k, v in plan_opts.items():
            plan.set(k, v)
        hub_plan = vq_plan = wav = None
        if e2e:
            from fdm_amd.hubert import WAV2VEC2_BASE, HubertPlan
            from fdm_amd.vq import VQPlan
            side_dt = dt if dt in (F32, DTYPE_NAMES["bf16"]) else F32     # VQ quant / decode run fp32 in the split modes (wider parity margin)
            hub_dt = dt if dt in (F32, DTYPE_NAMES["bf16"], DTYPE_NAMES["f16x3"]) else (DTYPE_NAMES["f16x3"] if dt == DTYPE_NAMES["f16"] else F32)      # HuBERT: fp32, bf16, or split-fp16 layers (also under the single-plane fp16 step program)
            if shipped and shipped[5] == "wav2vec":
                hub_plan = HubertPlan(W.make_wav2vec_weights(12), 12, hub_dt, dev, cfg=WAV2VEC2_BASE)
            else:
                hub_plan = HubertPlan(W.make_hubert_weights(24), 24, hub_dt, dev)
            vq_plan = VQPlan(preset, W.make_vq_weights(preset), side_dt, dev)
            # audio keyed by the GLOBAL clip index (like the Philox noise): a clip is the same clip on whichever rank it lands
            n_wav = int(shipped[1] * 16000) if shipped else 160000
            wav = (torch.cat([torch.randn(1, n_wav, generator=torch.Generator().manual_seed(100 + rank * B + b)) for b in range(B)]) * 0.1).to(dev)
        emo_b = emo                                          # [B*S] rows for the batched call (None where the preset has none)
        emo_clip = None if emo is None else emo[:B]          # one row per clip: the codebook slice of the EVQ quantiser

        def prep(marks=None):
            if e2e:
                feat = hub_plan.forward(wav)
                if marks is not None:
                    marks.append(torch.cuda.Event(enable_timing=True)); marks[-1].record()
                plan.prepare(feat[:, :L * p.pair], style, emo_b, L=L, n_conds=S)
            else:
                plan.prepare(hub, style, emo, L=L, cfg=cfg, n_conds=S)
        prep()
        plan.tune()        # plan-time work, outside the timed region: GEMM tile tuning for this shape (cached in the plan)

        def one_call(collect=True, marks=None):
            """marks (a list): HIP events after every stage of the call -- start, audio encoder, tables, chain, quant, decode"""
            def mark():
                if marks is not None:
                    marks.append(torch.cuda.Event(enable_timing=True)); marks[-1].record()
            mark()
            if e2e or S > 1:   # per-clip work inside the call: HuBERT + tables (cfg5, shipped_*); the clip's tables for its S conditions (cfg1x8)
                prep(marks)
            mark()
            if sampler == "ddpm":
                out = plan.sample_ddpm(xT, ts, seed=1234, clip0=rank * B * S)
            else:
                out = plan.sample_ddim(xT, T)
            mark()
            if e2e and not a.profile_steps:      # (counter profiles of the step graph end with the chain)
                zq = vq_plan.quant(out * (1.5 / 1024), None if emo_clip is None else emo_clip.repeat_interleave(S, dim=0))[0]
                mark()
                out = vq_plan.decode(zq)
                mark()
            return gather_clips(out, dist, sizes=[B * S] * world) if collect else out     # equal shards: no size exchange

        el, ev_ms, out = timed(one_call, steps, warmup)
        assert torch.isfinite(out).all()
        n_launch = steps * n_live
        fl = step_flops(p, B * S, L, cfg)
        step_ms = ev_ms / n_launch
        ach = fl / (step_ms * 1e-3) / 1e12
        leg = {"dtype": dtype_name, "value": round(world * B * S * L * steps / el, 3), "unit": "frames/s", "steps": steps,
               "ms_per_step": round(el / steps * 1e3, 3),
               "diffusion_steps_per_s": round(n_launch / (ev_ms * 1e-3), 1),
               "kernel_launches_per_diffusion_step": plan.get("launches_per_step"),
               "host_graph_launches_per_sample": plan.get("graph_launches"), "denoiser_steps_per_sample": n_live,
               "gemm_tiles": {k: v for k, v in plan.tiles.items() if v},      # plan-time choice per call site (FDM_TILE_*; others: heuristic)
               "roofline": {"bound": "mfma", "kernel": "denoiser step graph (one diffusion step, all kernels)",
                            "achieved": round(ach, 2), "peak": PEAK[dtype_name], "unit": "TFLOP/s",
                            "frac": round(ach / PEAK[dtype_name], 4), "traffic": None,
                            "flops_per_launch": fl, "avg_launch_ms": round(step_ms, 5)}}
        if shipped and not a.profile_steps:
            # per-stage split of the call (HIP events between the stages, three extra calls outside the timed region)
            acc = [0.0] * 5
            for _ in range(3):
                marks = []
                one_call(False, marks)
                torch.cuda.synchronize()
                for i in range(5):
                    acc[i] += marks[i].elapsed_time(marks[i + 1]) / 3.0
            leg["stages_ms"] = {"audio_encoder": round(acc[0], 3), "tables": round(acc[1], 3), f"chain_{n_live}_denoiser_calls": round(acc[2], 3),
                                "quant": round(acc[3], 3), "decode": round(acc[4], 3), "sum": round(sum(acc), 3)}
            # the step graph's roofline entry from the chain stage alone (the call also holds the once-per-clip stages)
            step_ms = acc[2] / n_live
            ach = fl / (step_ms * 1e-3) / 1e12
            leg["roofline"].update({"achieved": round(ach, 2), "frac": round(ach / PEAK[dtype_name], 4), "avg_launch_ms": round(step_ms, 5)})
            leg["diffusion_steps_per_s"] = round(n_live / (acc[2] * 1e-3), 1)
        if shipped and S > 1 and not a.profile_steps:
            # the reference's loop as written (samples/sample_diffusion_vocaset.py:71-88): per style one-hot, one B = 1 pipeline --
            # audio encoder, tables, DDIM chain, quant, decode -- with the same audio
            st1 = [style[i:i + 1] for i in range(B * S)]

            def seq_pipeline(collect=True):
                outs = []
                for r in range(B * S):
                    feat = hub_plan.forward(wav)
                    plan.prepare(feat[:, :L * p.pair], st1[r], emo_clip, L=L)
                    lat = plan.sample_ddim(xT[r:r + 1], T) if sampler == "ddim" else plan.sample_ddpm(xT[r:r + 1], ts, seed=1234, clip0=r)
                    outs.append(vq_plan.decode(vq_plan.quant(lat * (1.5 / 1024), emo_clip)[0]))
                o = torch.cat(outs)
                return gather_clips(o, dist, sizes=[B * S] * world) if collect else o
            plan.prepare(hub_plan.forward(wav)[:, :L * p.pair], st1[0], emo_clip, L=L)
            plan.tune()
            el_s, ev_s, out_s = timed(seq_pipeline, steps, warmup)
            leg["sequential_loop"] = {"value": round(world * B * S * L * steps / el_s, 3), "unit": "frames/s",
                                      "ms_per_step": round(el_s / steps * 1e3, 3),
                                      "what": f"{S} sequential B = 1 pipelines per clip (audio encoder + tables + chain + quant + decode each: "
                                              "samples/sample_diffusion_vocaset.py:71-88)",
                                      "bit_identical_to_batched": bool(torch.equal(out_s, out))}
            leg["speedup_vs_sequential_loop"] = round(el_s / el, 3)
            prep()
        elif S > 1 and not a.profile_steps:
            # the reference's own loop: one B = 1 sampling call per style, same audio (tables rebuilt per call, as a caller that
            # loops ddim_sample does); same plan, same tiles policy, same timing harness
            st1 = [style[i:i + 1] for i in range(B * S)]

            def seq_call(collect=True):
                outs = []
                for b in range(B):
                    for s_ in range(S):
                        r = b * S + s_
                        plan.prepare(hub[b:b + 1], st1[r], None if emo is None else emo[b:b + 1], L=L, cfg=cfg)
                        outs.append(plan.sample_ddim(xT[r:r + 1], T) if sampler == "ddim" else plan.sample_ddpm(xT[r:r + 1], ts, seed=1234, clip0=rank * B * S + r))
                o = torch.cat(outs)
                return gather_clips(o, dist, sizes=[B * S] * world) if collect else o
            plan.prepare(hub[:1], st1[0], None if emo is None else emo[:1], L=L, cfg=cfg)
            plan.tune()
            el_s, ev_s, out_s = timed(seq_call, steps, warmup)
            leg["sequential_loop"] = {"value": round(world * B * S * L * steps / el_s, 3), "unit": "frames/s",
                                      "ms_per_step": round(el_s / steps * 1e3, 3),
                                      "what": f"{S} sequential B = 1 calls per clip (samples/sample_diffusion_vocaset.py:71-83)",
                                      "bit_identical_to_batched": bool(torch.equal(out_s, out))}
            leg["speedup_vs_sequential_loop"] = round(el_s / el, 3)
        if world > 1 and rank == 0 and not e2e and S == 1 and not a.profile_steps:
            # N > 1 self-check, outside the timed region: rank 0 recomputes the LAST rank's clips (same global clip indices, so the
            # same Philox streams) and compares them with what the collective delivered -- cross-rank determinism and the gather
            # order, verified on the hardware the line was measured on
            r = world - 1
            slr = slice(r * B, (r + 1) * B)
            hub_r = inp["hub"][slr][:, :, :768].contiguous() if preset == "biwi" else inp["hub"][slr]
            plan.prepare(hub_r, inp["style"][slr], inp["emo"][slr] if "emo" in inp else None, L=L, cfg=cfg)
            xr = inp["x"][slr].to(dev)
            mine = plan.sample_ddpm(xr, ts, seed=1234, clip0=r * B) if sampler == "ddpm" else plan.sample_ddim(xr, T)
            leg["shard_check"] = {"rank": r, "bit_identical": bool(torch.equal(mine.to(out.device), out[slr]))}
            prep()
        if want_parity and preset == "vocaset" and rank == 0:
            leg["parity_max_abs"] = parity_vs_reference(plan, dev)
        return leg, plan, out

    head, plan, out = run_leg(a.dtype, a.steps, a.warmup, not a.headline_only and not a.profile_steps)
    if a.dump and rank == 0:
        import numpy as np
        np.save(a.dump, out.float().cpu().numpy())
    contract = None
    if not a.headline_only and not a.profile_steps and a.dtype not in ("f32", "f16x3"):
        del plan
        contract, _, _ = run_leg("f16x3", a.contract_steps, a.warmup, True)
        contract["roofline"]["kernel"] = "denoiser step graph in the contract mode (split-fp16 operands, three 16-bit MFMA passes per product)"
    f16leg = None
    if not a.headline_only and not a.profile_steps and a.dtype == "bf16" and world == 1:
        # the throughput mode's closer twin (round 6): single-plane fp16 operands -- bf16's bytes and MFMA rate, 11 significand bits --
        # timed and checked against the same reference chains, so the line shows what the headline's 8 bits cost beside what they buy
        f16leg, _, _ = run_leg("f16", a.contract_steps, a.warmup, True)

    if rank == 0:
        # Counter-derived fields (HBM-side bytes per launch of the step graph, MFMA-busy share, the dominant kernel's own
        # roofline entry) come from the committed rocprofv3 --pmc summary of THIS configuration and mode (bench.py cannot
        # collect PMC counters on itself).  They are refused (null) when the summary was taken on another step program (its
        # kernel-launch count per diffusion step must equal the running build's); the tile set the summary was profiled with is
        # reported next to this run's (`counters_tiles`, `counters_tiles_match`), and the dominant kernel's entry is dropped when
        # this run's tiles at ITS call sites (out-proj, FFN2) differ from the profiled ones -- nothing is paired silently.
        def parse_tiles(txt):
            out = {}
            for part in str(txt or "").split(","):
                if "=" in part:
                    k, v = part.split("=", 1)
                    try:
                        out[k.strip()] = int(v)
                    except ValueError:
                        pass
            return out

        def counters(leg):
            r = leg["roofline"]
            r.update({"mfma_busy": None, "dominant_kernel": None, "counters_from": None, "counters_commit": None, "counters_date": None,
                      "counters_tiles": None, "counters_tiles_match": None})
            if a.batch:
                return
            for rd in ("r6", "r5", "r4", "r3", "r2"):
                try:
                    rel = f"profiles/{rd}_pmc_{a.config}_{leg['dtype']}/summary.json"
                    pm = json.load(open(os.path.join(ROOT, rel)))
                    if int(round(pm["summary"]["launches_per_step"])) != int(leg["kernel_launches_per_diffusion_step"]):
                        continue
                    prof_tiles = {k: v for k, v in parse_tiles(pm["summary"].get("tiles")).items() if v}
                    r["traffic"] = pm["summary"]["traffic_bytes_per_step"]
                    r["mfma_busy"] = pm["summary"]["mfma_busy_time_weighted"]
                    r["counters_from"] = rel
                    r["counters_commit"], r["counters_date"] = pm["summary"].get("commit"), pm["summary"].get("date")      # build and day the profile was taken on
                    r["counters_tiles"] = prof_tiles
                    r["counters_tiles_match"] = prof_tiles == leg["gemm_tiles"]
                    leg["roofline_hbm"] = hbm_class_roofline(pm, rel, p, B * S * L * (2 if cfg else 1), leg["dtype"])
                    ks = [k for k in pm["kernels"] if "gemm" in k["kernel"]]
                    same_sites = all(prof_tiles.get(site, 0) == leg["gemm_tiles"].get(site, 0) for site in ("out", "ffn2"))
                    if ks and same_sites:
                        k = max(ks, key=lambda q: q["launches_per_step"] * q["avg_us"])
                        # its own roofline entry: the kernel serves the out-proj (K = d) and FFN2 (K = ffn) sites, rows x d outputs each
                        rows_k = B * S * L * (2 if cfg else 1)
                        k_fl = 2.0 * rows_k * p.d * (p.d + p.ffn) / 2.0 if k["launches_per_step"] == 2 * p.n_layers else None
                        k_ach = k_fl / (k["avg_us"] * 1e-6) / 1e12 if k_fl else None
                        r["dominant_kernel"] = {"name": k["kernel"], "launches_per_step": k["launches_per_step"], "avg_us_profiled": k["avg_us"],
                                                "flops_per_launch": k_fl, "achieved": round(k_ach, 1) if k_ach else None, "unit": "TFLOP/s",
                                                "frac": round(k_ach / PEAK[leg["dtype"]], 4) if k_ach else None,
                                                "mfma_busy": k.get("mfma_busy"), "wave_cycles_waiting": k.get("wait"), "l2_hit": k.get("l2_hit"),
                                                "fetch_mb": k.get("fetch_mb"), "write_mb": k.get("write_mb")}
                    return
                except (OSError, ValueError, KeyError):
                    continue
        counters(head)
        res = {
            "metric": "animated frames/sec (1000-step DDPM, VOCASET FDM) at 1/2/4/8 MI355X",
            "value": head["value"], "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{a.config}: {preset} FDM, {B} clips/GPU x {L} latent frames"
                                   + (f" x {S} style conditions per clip in one step program" if S > 1 else "")
                                   + f", {T}-step {sampler.upper()}{' + CFG 2.5' if cfg else ''}, random-init weights, "
                                   + ((f"{shipped[1]:g} s synthetic audio -> {'wav2vec2-base' if shipped[5] == 'wav2vec' else 'HuBERT-large'} -> tables -> sample -> "
                                       f"{'EVQ ' if p.n_emo else ''}quant -> decode to {p.V3 // 3}-vertex meshes, all inside the timed call (the reference's "
                                       f"samples/sample_diffusion_{'vocaset' if preset == 'vocaset' else preset}.py per test clip)") if shipped else
                                      "10 s synthetic audio -> HuBERT-large -> sample -> VQ quant + decode to 5023-vertex meshes"
                                      if e2e else "synthetic audio-encoder features, Philox noise"), "global_batch": B * world,
                       "latent_frames": L, "diffusion_steps": T, "parallelism": f"clip-shard x{world}"},
            "rccl_ranks": (dist.get_world_size() if dist is not None else 1),
            "weights": ("broadcast from rank 0 (one flattened collective)" if (a.broadcast_weights and dist is not None) else "generated per rank from the seed"),
            "dist_backend": (dist.get_backend() if dist is not None else None),
            "diffusion_steps_per_s": head["diffusion_steps_per_s"],
            "kernel_launches_per_diffusion_step": head["kernel_launches_per_diffusion_step"],
            "host_graph_launches_per_sample": head["host_graph_launches_per_sample"],
            "denoiser_steps_per_sample": head["denoiser_steps_per_sample"],
            "gemm_tiles": head["gemm_tiles"],
            "plan_options": plan_opts,
            "roofline": head["roofline"],
        }
        for k in ("roofline_hbm", "stages_ms", "sequential_loop", "speedup_vs_sequential_loop", "shard_check"):
            if k in head:
                res[k] = head[k]
        if "parity_max_abs" in head:
            res["parity"] = {"dtype": a.dtype, "max_abs": head["parity_max_abs"], "tolerance": PARITY_TOL[a.dtype],
                             "within_1e-4_contract": head["parity_max_abs"] < 1e-4,
                             "against": "tests/golden/chains_vocaset.npz (reference outputs: DDPM t=9..0 and 999..990 after every step, DDIM 50)"}
        if contract is not None:
            counters(contract)
            res["contract_mode"] = {"dtype": "f16x3", "value": contract["value"], "unit": "frames/s", "steps": contract["steps"],
                                    "ms_per_step": contract["ms_per_step"], "roofline": contract["roofline"],
                                    "gemm_tiles": contract["gemm_tiles"],
                                    "parity_max_abs": contract.get("parity_max_abs"), "tolerance": 1e-4,
                                    "what": "the same workload in the arithmetic mode that meets north_star's 1e-4 max-abs tolerance"}
        if f16leg is not None:
            res["f16_mode"] = {"dtype": "f16", "value": f16leg["value"], "unit": "frames/s", "steps": f16leg["steps"], "ms_per_step": f16leg["ms_per_step"],
                               "roofline_frac": f16leg["roofline"]["frac"], "gemm_tiles": f16leg["gemm_tiles"],
                               "parity_max_abs": f16leg.get("parity_max_abs"), "tolerance": PARITY_TOL["f16"], "within_1e-4_contract": False,
                               "what": "the same workload on single-plane fp16 operands (FDM_F16: the split kind's hi plane alone)"}
        if not a.no_cpu_baseline and world == 1:
            # the bounded CPU sample is defined for the denoiser-only configs; cfg4/cfg5 reuse cfg2's shape class
            res["cpu_baseline"] = cpu_baseline(a.config if a.config in ("cfg1", "cfg2", "cfg3") else ("cfg1" if (a.config == "cfg1x8" or shipped) else "cfg2"))
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
