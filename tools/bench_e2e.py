"""Phase timings of the end-to-end clip pipeline on one MI355X (cfg5 per-GPU shape: B clips x 10 s audio ->
HuBERT once -> 1000-step DDPM -> quant -> decode)."""
import sys, time, torch
sys.path.insert(0, 'face-diffusion-model_amd')
from fdm_amd import synth
from fdm_amd._lib import BF16, F16X3, F32
from fdm_amd.denoiser import DenoiserPlan
from fdm_amd.hubert import HubertPlan, num_frames
from fdm_amd.vq import VQPlan
DEV = 'cuda:0'
dt = {'bf16': BF16, 'f16x3': F16X3}.get(sys.argv[1] if len(sys.argv) > 1 else 'bf16', F32)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
T = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
def sync(): torch.cuda.synchronize()
def timed(name, fn, reps=1):
    sync(); t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    sync(); t = (time.perf_counter() - t0) / reps
    print(f"{name:28s} {t*1e3:10.2f} ms")
    return out, t
hub_plan = HubertPlan(synth.make_hubert_weights(24), 24, dt, DEV)
den = DenoiserPlan('vocaset', synth.make_fdm_weights('vocaset'), dt, DEV)
vq = VQPlan('vocaset', synth.make_vq_weights('vocaset'), F32 if (dt == F16X3 and 'vq16' not in sys.argv) else dt, DEV)
n = int(secs * 16000)
g = torch.Generator().manual_seed(0)
wav = (torch.randn(B, n, generator=g) * 0.1).to(DEV)
L = min(num_frames(n), 600)
hub_plan.forward(wav)                         # warm-up
hub, t_h = timed('hubert (once per clip)', lambda: hub_plan.forward(wav))
sty = torch.eye(8)[torch.arange(B) % 8]
_, t_p = timed('prepare (AF, C1 tables)', lambda: den.prepare(hub, sty, L=L))
xT = torch.randn(B, L * 16, 64, device=DEV)
ts = list(range(T - 1, -1, -1))
den.tune()
den.sample_ddpm(xT, ts[:5], seed=1)
lat, t_s = timed(f'sample {T} DDPM steps', lambda: den.sample_ddpm(xT, ts, seed=1))
lat = lat * (1.5 / 256 / 4)
vq.decode(vq.quant(lat)[0])
(zq, idx), t_q = timed('vq quant', lambda: vq.quant(lat))
out, t_d = timed('vq decode', lambda: vq.decode(zq))
tot = t_h + t_p + t_s + t_q + t_d
fl_h = 2 * B * L * 24 * (4 * 1024 * 1024 + 2 * 1024 * 4096) + 24 * B * 4 * L * L * 1024
print(f"B={B} L={L} dtype={ {BF16: 'bf16', F16X3: 'f16x3'}.get(dt, 'f32') }: end-to-end {tot*1e3:.1f} ms -> {B*L/tot:.1f} frames/s "
      f"(sampling share {100*t_s/tot:.1f}%); hubert encoder layers ~{fl_h/t_h/1e12:.0f} TFLOP/s")
