"""Helper of tests/test_hubert_vq_gpu.py::test_encoder_is_reproducible_when_another_process_shares_the_gpu: runs HuBERT-large forwards
on cuda:0 and prints how many distinct outputs it saw.   python corun_encoder.py <seconds or 0> <forwards> <dtype>"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "face-diffusion-model_amd"))
from fdm_amd import ops, synth  # noqa: E402
from fdm_amd._lib import DTYPE_NAMES  # noqa: E402
from fdm_amd.hubert import HubertPlan  # noqa: E402

seconds, forwards, dtype = float(sys.argv[1]), int(sys.argv[2]), DTYPE_NAMES[sys.argv[3]]
DEV = "cuda:0"
wav = (torch.cat([torch.randn(1, 160000, generator=torch.Generator().manual_seed(100 + b)) for b in range(4)]) * 0.1).to(DEV)
hub = HubertPlan(synth.make_hubert_weights(2), 2, dtype, DEV)
if seconds > 0:                       # the co-runner: just keep the device busy with the same program
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(10):
            hub.forward(wav)
        torch.cuda.synchronize()
    sys.exit(0)
# conv 0 + LayerNorm + GELU alone (the kernel that was not reproducible), then the encoder
n, B = 160000, 4
T0 = (n - 10) // 5 + 1
g = torch.Generator().manual_seed(1)
w0, b0 = (torch.randn(512, 10, generator=g) * 0.3).to(DEV), (torch.randn(512, generator=g) * 0.1).to(DEV)
g0, be0 = torch.randn(512, generator=g).to(DEV), torch.randn(512, generator=g).to(DEV)
x = torch.zeros(B * T0, 512, device=DEV)
sums = set()
for _ in range(forwards):
    ops.conv0_ln_gelu(wav, w0, b0, g0, be0, x, B, n, T0)
    sums.add(float(x.double().sum()))
outs = set()
for _ in range(forwards):
    outs.add(float(hub.forward(wav).double().sum()))
print(f"distinct conv0 outputs {len(sums)} distinct encoder outputs {len(outs)}")
