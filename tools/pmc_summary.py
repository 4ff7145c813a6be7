"""Summarise a `rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv` run of bench.py: per-kernel averages and the
per-diffusion-step total over the LAST n_steps * launches_per_step fdm kernels (the step graph replays; earlier rows
are table building and plan-time tuning).  usage: pmc_summary.py counter_collection.csv out.csv [n_steps] [launches]"""
import collections
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
per_step = int(sys.argv[4]) if len(sys.argv) > 4 else 51
rows = [r for r in csv.DictReader(open(src)) if "fdm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
tail = rows[-n_steps * per_step:]
acc = collections.OrderedDict()
for r in tail:
    k = (r["Kernel_Name"], r["Grid_Size"], r["Workgroup_Size"])
    a = acc.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
name = tail[0]["Counter_Name"]
with open(dst, "w") as f:
    f.write(f"kernel,grid_size,workgroup_size,dispatches,avg_{name}_KB,total_KB\n")
    for (k, g, w), (n, v) in acc.items():
        f.write(f"\"{k}\",{g},{w},{n},{v / n:.1f},{v:.1f}\n")
    tot = sum(v for _, v in acc.values())
    f.write(f"\"TOTAL per diffusion step ({n_steps} steps x {per_step} launches)\",,,{len(tail)},{tot / n_steps:.1f},{tot:.1f}\n")
print(f"{name}: {tot / n_steps / 1024:.2f} MB per diffusion step as reported")
