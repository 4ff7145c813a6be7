"""The once-per-clip stages alone, for profiling (round 4, review item 2): HuBERT-large / wav2vec2-base forward and the VQ decoder.

    python tools/bench_encoders.py <stage> <mode> <B> <seconds> [reps]
      stage: hubert | wav2vec | vqdecode      mode: bf16 | f16x3 | f32
Prints wall time per call (events around `reps` calls after a warm-up) and the algorithmic TFLOP/s of the transformer layers.
Under `rocprofv3 --kernel-trace --stats` / `--pmc` (tools/profile_encoders.sh) every launch of the process belongs to the stage."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'face-diffusion-model_amd'))
from fdm_amd import synth
from fdm_amd._lib import BF16, F16X3, F32
from fdm_amd.hubert import HUBERT_LARGE, WAV2VEC2_BASE, HubertPlan, num_frames
from fdm_amd.vq import VQPlan

DEV = 'cuda:0'
stage, mode = sys.argv[1], sys.argv[2]
B, secs = int(sys.argv[3]), float(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dt = {'bf16': BF16, 'f16x3': F16X3, 'f32': F32}[mode]
n = int(secs * 16000)
g = torch.Generator().manual_seed(0)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if stage in ('hubert', 'wav2vec'):
    large = stage == 'hubert'
    nl, D, FFN = (24, 1024, 4096) if large else (12, 768, 3072)
    plan = HubertPlan(synth.make_hubert_weights(nl) if large else synth.make_wav2vec_weights(nl), nl, dt, DEV, cfg=HUBERT_LARGE if large else WAV2VEC2_BASE)
    wav = (torch.randn(B, n, generator=g) * 0.1).to(DEV)
    N = num_frames(n)
    ms = timed(lambda: plan.forward(wav))
    fl = 2.0 * B * N * nl * (4 * D * D + 2 * D * FFN) + nl * B * 4.0 * N * N * D
    conv = 2.0 * B * sum(t * 512 * 512 * k for t, k in zip([n // 10, n // 20, n // 40, n // 80, n // 160, n // 320], [3, 3, 3, 3, 2, 2]))
    print(f"{stage} {mode} B={B} {secs:g} s audio -> {N} frames: {ms:.3f} ms per call; encoder layers {fl / 1e9:.0f} GFLOP (+ conv stack ~{conv / 1e9:.0f}) "
          f"-> {(fl + conv) / ms / 1e9:.0f} TFLOP/s = {(fl + conv) / ms / 1e9 / (157.3 if mode == 'f32' else 2500) * 100:.1f} % of the {'fp32' if mode == 'f32' else '16-bit'} MFMA peak")
else:
    L = min(num_frames(n), 600)
    vq = VQPlan('vocaset', synth.make_vq_weights('vocaset'), dt, DEV)
    lat = torch.randn(B, L * 16, 64, generator=g).to(DEV) * 0.01
    zq, _ = vq.quant(lat)
    ms = timed(lambda: vq.decode(zq))
    fl = B * L * 131.6e6
    print(f"vqdecode {mode} B={B} L={L}: {ms:.3f} ms per call; {fl / 1e9:.0f} GFLOP -> {fl / ms / 1e9:.0f} TFLOP/s")
