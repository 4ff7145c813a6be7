"""summary.md of one tools/profile_encoders.sh directory: per (kernel, grid) class of a once-per-clip stage -- launches per call,
average duration, share of the call, MFMA-busy, waiting share, L2 hit, fetched / written MB per launch.
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES / 32 shader engines); wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES
  fetch MB = FETCH_SIZE (KB) x 2 / 1024 (gfx950 tallies 16-B-per-lane streams at 1/2: MI355X_MICROARCH.md, HBM); write MB = WRITE_SIZE / 1024
usage: encoder_report.py <dir> "<title>" """
import collections
import csv
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_report import short  # noqa: E402

d, title = sys.argv[1], sys.argv[2]
wall = open(os.path.join(d, "wall.txt")).read().strip().splitlines()[-1] if os.path.exists(os.path.join(d, "wall.txt")) else ""
rows = [r for r in csv.DictReader(open(os.path.join(d, "trace.csv"))) if "fdm" in r["Kernel_Name"]]
tlog = open(os.path.join(d, "trace.log")).read() if os.path.exists(os.path.join(d, "trace.log")) else ""
# calls of the stage in the traced process = 3 warm-up + reps (bench_encoders.py); reps is the last CLI argument echoed nowhere, so
# take the most common launch count of the layer kernels: every per-layer kernel runs n_layers x calls times
cnt = collections.Counter()
dur = collections.Counter()
for r in rows:
    # (the counter CSVs carry the total grid / workgroup sizes, the trace one value per dimension)
    gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))
    wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0))) * int(r.get("Workgroup_Size_Y", 1)) * int(r.get("Workgroup_Size_Z", 1))
    k = (short(r["Kernel_Name"]), str(gx), str(wx))
    cnt[k] += 1
    dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
n_calls = int(os.environ.get("ENC_REPS", "10")) + 3


def counters(name):
    path = os.path.join(d, name + ".raw.csv")
    acc = collections.defaultdict(lambda: collections.Counter())
    n = collections.Counter()
    if not os.path.exists(path):
        return acc, n
    seen = set()
    for r in csv.DictReader(open(path)):
        if "fdm" not in r["Kernel_Name"]:
            continue
        k = (short(r["Kernel_Name"]), r["Grid_Size"], r["Workgroup_Size"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], k) not in seen:
            seen.add((r["Dispatch_Id"], k)); n[k] += 1
    return acc, n


sq, nsq = counters("sq")
fe, nfe = counters("fetch")
wr, nwr = counters("write")
tot_us = sum(dur.values()) / 1e3 / n_calls
print(f"# {title}\n\n{wall}\n\nkernel time per call {tot_us:.1f} us over {sum(cnt.values()) / n_calls:.0f} launches ({n_calls} calls traced; one-time weight preparation shows as < 1 launch per call)\n")
cols = ["kernel", "grid", "wg", "launches/call", "avg us", "us/call", "share", "mfma_busy", "wait", "l2_hit", "fetch MB", "write MB"]
print("| " + " | ".join(cols) + " |\n|" + "---|" * len(cols))
tw = 0.0
for k, ns in sorted(dur.items(), key=lambda kv: -kv[1]):
    n = cnt[k]
    avg = ns / n / 1e3
    o = [re.sub(r"\s+", " ", k[0])[:110], k[1], k[2], f"{n / n_calls:.2f}", f"{avg:.2f}", f"{ns / 1e3 / n_calls:.1f}", f"{100 * ns / 1e3 / n_calls / tot_us:.1f} %"]
    mf = wt = hit = fm = wm = ""
    if nsq[k]:
        c = sq[k]
        cyc = c["SQ_BUSY_CYCLES"] / 32.0
        if cyc:
            mfv = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
            mf = f"{100 * mfv:.1f} %"
            tw += mfv * ns
        if c["SQ_WAVE_CYCLES"]:
            wt = f"{100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.0f} %"
    if nfe[k]:
        c = fe[k]
        fm = f"{c['FETCH_SIZE'] * 2 / 1024.0 / nfe[k]:.2f}"
        miss = wr[k]["TCC_MISS_sum"] / nwr[k] if nwr[k] else 0.0
        hits = c["TCC_HIT_sum"] / nfe[k]
        if hits + miss:
            hit = f"{100 * hits / (hits + miss):.0f} %"
    if nwr[k]:
        wm = f"{wr[k]['WRITE_SIZE'] / 1024.0 / nwr[k]:.2f}"
    print("| " + " | ".join(o + [mf, wt, hit, fm, wm]) + " |")
print(f"\ntime-weighted MFMA busy over the call: {100 * tw / max(sum(dur.values()), 1):.1f} %")
