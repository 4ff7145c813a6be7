"""Where a GEMM's time goes inside a dependent chain: NL GEMMs back to back (each reads the previous one's output, distinct
weights), replayed as one hipGraph; every workgroup stamps the 100 MHz clock at entry, after its first k-tile landed, after the
k loop, after its stores were issued and after they were acknowledged.  Needs the instrumented library
(`make -C face-diffusion-model_amd/csrc stamps`, then FDM_LIB_PATH=.../libfdm_hip_stamps.so; the stamp buffer rides in
fdm_gemm_args.incr_table).  usage: gemm_timeline.py [dtype] [M] [N=K] [tile]"""
import sys
sys.path.insert(0, 'face-diffusion-model_amd')
import torch
from fdm_amd import ops
from fdm_amd._lib import DTYPE_NAMES, F32
DEV = 'cuda:0'
dt = DTYPE_NAMES[sys.argv[1]] if len(sys.argv) > 1 else 1
M = int(sys.argv[2]) if len(sys.argv) > 2 else 800
N = K = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
tile = int(sys.argv[4]) if len(sys.argv) > 4 else 1
BM, BN = {1: (64, 64), 2: (128, 64), 3: (128, 128), 8: (64, 64), 6: (64, 64), 9: (32, 64)}[tile]
NL = 12
g = torch.Generator().manual_seed(0)
mk = lambda r, c: ops.to_operand((torch.randn(r, c, generator=g) * 0.03).to(DEV), dt)
Ws = [mk(N, K) for _ in range(NL)]
bufs = [mk(M, K), mk(M, K)]
bias = torch.zeros(N, device=DEV)
nwg = -(-M // BM) * (N // BN)
stamps = torch.zeros(NL, nwg, 8, dtype=torch.int64, device=DEV)
prog = ops.Program()
with prog:
    for l in range(NL):
        ops.gemm(bufs[l % 2], Ws[l], M, N, K, bias=bias, out_t=bufs[(l + 1) % 2], tile=tile, incr_table=stamps[l])
prog.instantiate()
prog.replay(5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); prog.replay(20); e1.record(); e1.synchronize()
print(f"{sys.argv[1:]}: {nwg} workgroups, {e0.elapsed_time(e1) / 20 / NL * 1e3:.2f} us per GEMM by events")
s = stamps.cpu().double() * 0.01      # us
rows, first_wave, kernarg = [], [], []
for l in range(2, NL):
    t0, t1, t2, t3, t4 = (s[l, :, i] for i in range(5))
    tf = s[l, :, 5]
    prev_end = s[l - 1, :, 4].max()
    first_wave.append(float(tf.min() - prev_end)); kernarg.append(float((t0 - tf).median()))
    rows.append([float(x) for x in (t0.min() - prev_end, t0.max() - t0.min(), (t1 - t0).median(), (t2 - t1).median(), (t3 - t2).median(),
                                    (t4 - t3).median(), t4.max() - t0.min(), t4.max() - t4.median())])
r = torch.tensor(rows).mean(0)
print("  previous kernel's last ack -> first workgroup entry : %.2f us" % r[0])
print("     of which: last ack -> first instruction of the first wave %.2f us; first instruction -> kernel arguments read %.2f us (median)" % (sum(first_wave) / len(first_wave), sum(kernarg) / len(kernarg)))
print("  entry spread over workgroups                         : %.2f us" % r[1])
print("  entry -> first k-tile landed (median workgroup)      : %.2f us" % r[2])
print("  k loop                                               : %.2f us" % r[3])
print("  epilogue until stores issued                         : %.2f us" % r[4])
print("  stores acknowledged                                  : %.2f us" % r[5])
print("  first entry -> last ack (kernel span)                : %.2f us;  last ack - median ack %.2f us" % (r[6], r[7]))
