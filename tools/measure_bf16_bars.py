"""Measured distances of the bf16 side stages against the reference's golden vectors (run on the GPU box): the stated
bars in tests/ are 2x these.  HuBERT-large 24 layers 2 s, wav2vec2-base 12 layers 2 s, VQ decode (VOCASET, 7 frames x 3 clips)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-diffusion-model_amd")]
import numpy as np, torch
from fdm_amd._lib import BF16, F32
from fdm_amd.hubert import HubertPlan, WAV2VEC2_BASE
from fdm_amd.vq import VQPlan
from oracle import hubert_oracle as HO, vq_oracle as VO, weights as W
DEV = "cuda:0"
mad = lambda a, b: float((torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max())
g = np.load(os.path.join(ROOT, "tests/golden/hubert.npz"))
wav = HO.processor_normalize(torch.randn(32000, generator=torch.Generator().manual_seed(12)) * 0.1)
print("hubert-large bf16 2 s vs golden:", mad(HubertPlan(W.make_hubert_weights(24), 24, BF16, DEV).forward(wav)[0], g["out_L24_2s"]), "|out|max", float(np.abs(g["out_L24_2s"]).max()))
g2 = np.load(os.path.join(ROOT, "tests/golden/wav2vec.npz"))
wv = HO.processor_normalize(torch.randn(32000, generator=torch.Generator().manual_seed(22)) * 0.1)
print("wav2vec2-base bf16 2 s vs golden:", mad(HubertPlan(W.make_wav2vec_weights(12), 12, BF16, DEV, cfg=WAV2VEC2_BASE).forward(wv)[0], g2["out_L12_2s"]), "|out|max", float(np.abs(g2["out_L12_2s"]).max()))
w = W.make_vq_weights("vocaset")
z = torch.randn(3, 7 * 16, 64, generator=torch.Generator().manual_seed(9)) * (1.5 / 256)
zq, _ = VO.quant(w, "vocaset", z)
ref = VO.decode(w, "vocaset", zq)
print("vq decode bf16 vs oracle:", mad(VQPlan("vocaset", w, BF16, DEV).decode(zq.to(DEV)), ref), "|out|max", float(ref.abs().max()))
