"""CPU emulation of the split-operand GEMM modes (tools only; imports the oracle):
max-abs error of one full-size VOCASET denoiser call vs an fp64 run of the same network, with every nn.Linear evaluated as
  f32     plain fp32 (what the F32 mode does)
  bf16    operands rounded to bf16 (BF16 mode)
  bf16x3  operands split hi+lo in bf16, products hi.hi + hi.lo + lo.hi
  f16x3   operands split hi+lo' in fp16 (lo' = residual * 2^11), same three products
  bf16x6  three-way bf16 split, six products
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import fdm_oracle as FO
from oracle import weights as W

real_linear = F.linear

def split2(x, dt, scale):
    hi = x.to(dt).float()
    lo = ((x - hi) * scale).to(dt).float()
    return hi, lo

def make_linear(mode):
    def lin(x, w, b=None):
        if mode == "f32" or x.dim() < 2 or x.shape[-1] < 64:
            return real_linear(x, w, b)
        if mode == "f64":
            return real_linear(x.double(), w.double(), None if b is None else b.double())
        if mode == "bf16":
            y = real_linear(x.bfloat16().float(), w.bfloat16().float())
        elif mode in ("bf16x3", "f16x3"):
            dt, sc = (torch.bfloat16, 1.0) if mode == "bf16x3" else (torch.float16, 2048.0)
            xh, xl = split2(x, dt, sc)
            wh, wl = split2(w, dt, sc)
            y = real_linear(xh, wh) + (real_linear(xh, wl) + real_linear(xl, wh)) / sc
        elif mode == "bf16x6":
            x1 = x.bfloat16().float(); r = x - x1; x2 = r.bfloat16().float(); x3 = (r - x2).bfloat16().float()
            w1 = w.bfloat16().float(); r = w - w1; w2 = r.bfloat16().float(); w3 = (r - w2).bfloat16().float()
            y = real_linear(x1, w1) + (real_linear(x1, w2) + real_linear(x2, w1)) + (real_linear(x2, w2) + real_linear(x1, w3) + real_linear(x3, w1))
        return y if b is None else y + b
    return lin

def run(mode, w, preset, inp, t):
    FO.F.linear = make_linear(mode)
    try:
        if mode == "f64":
            torch.set_default_dtype(torch.float64)
            w64 = {k: v.double() for k, v in w.items()}
            bm = FO.biased_mask
            FO.biased_mask = lambda *a: bm(*a).double()
            out = FO.fdm_forward(w64, preset, inp["hub"].double(), t, inp["x"].double(), inp["style"].double(),
                                 None if "emo" not in inp else inp["emo"].double(), folded=True)
            torch.set_default_dtype(torch.float32)
            return out
        return FO.fdm_forward(w, preset, inp["hub"], t, inp["x"], inp["style"], inp.get("emo"), folded=True)
    finally:
        FO.F.linear = real_linear
        if mode == "f64":
            FO.biased_mask = bm
        torch.set_default_dtype(torch.float32)

if __name__ == "__main__":
    preset = sys.argv[1] if len(sys.argv) > 1 else "vocaset"
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    torch.set_num_threads(8)
    w = W.make_fdm_weights(preset)
    inp = W.synth_inputs(preset, 1, L, seed=3)
    for t in (999, 500, 1):
        ref = run("f64", w, preset, inp, t)
        print(f"{preset} L={L} t={t}: |x0|max = {float(ref.abs().max()):.3f}")
        for mode in ("f32", "f16x3", "bf16x6", "bf16x3", "bf16"):
            out = run(mode, w, preset, inp, t)
            print(f"   {mode:7s} max-abs err vs fp64 = {float((out.double() - ref).abs().max()):.3e}")
