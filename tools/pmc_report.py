"""Reduce rocprofv3 CSVs of a bench.py run to per-kernel rows of the timed step graph.

  pmc_report.py trace  kernel_trace.csv        out.csv n_steps launches_per_step   -> per-kernel launches/step, avg duration
  pmc_report.py reduce counter_collection.csv  out.csv n_steps launches_per_step   -> per-kernel average of every counter
  pmc_report.py merge  <dir> out.json [peak_tflops flops_per_step]                 -> one table from trace.csv + pass CSVs

Only the LAST n_steps * launches_per_step fdm kernels are used: the replays of the step graph (earlier rows are table
building and plan-time tuning)."""
import collections
import csv
import json
import os
import sys


def short(name):
    n = name.replace("void fdm::", "").replace("fdm::", "")
    return n.split("(")[0]


def tail_rows(rows, n_steps, per_step, order_key):
    rows = [r for r in rows if "fdm" in r["Kernel_Name"]]
    rows.sort(key=order_key)
    return rows[-n_steps * per_step:]


def cmd_trace(src, dst, n_steps, per_step):
    rows = list(csv.DictReader(open(src)))
    tail = tail_rows(rows, n_steps, per_step, lambda r: int(r["Start_Timestamp"]))
    acc = collections.OrderedDict()
    for r in tail:
        g = r.get("Grid_Size_X", r.get("Grid_Size", ""))
        a = acc.setdefault((short(r["Kernel_Name"]), g, r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))), [0, 0])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    span = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / n_steps / 1e3
    with open(dst, "w") as f:
        f.write("kernel,grid,workgroup,launches_per_step,avg_us,us_per_step\n")
        for (k, g, w), (n, ns) in acc.items():
            f.write(f"\"{k}\",{g},{w},{n / n_steps:.2f},{ns / n / 1e3:.3f},{ns / n_steps / 1e3:.2f}\n")
        tot = sum(ns for _, ns in acc.values()) / n_steps / 1e3
        f.write(f"\"TOTAL kernel time per diffusion step\",,,{len(tail) / n_steps:.1f},,{tot:.2f}\n")
        f.write(f"\"wall span per diffusion step (first start to last end)\",,,,,{span:.2f}\n")
    print(f"trace: {tot:.1f} us of kernels in a {span:.1f} us step ({len(tail) // n_steps} launches)")


def cmd_reduce(src, dst, n_steps, per_step):
    rows = list(csv.DictReader(open(src)))
    names = sorted({r["Counter_Name"] for r in rows})
    by_disp = collections.OrderedDict()
    for r in rows:
        if "fdm" not in r["Kernel_Name"]:
            continue
        d = by_disp.setdefault(int(r["Dispatch_Id"]), {"k": (short(r["Kernel_Name"]), r["Grid_Size"], r["Workgroup_Size"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    disp = [by_disp[k] for k in sorted(by_disp)][-n_steps * per_step:]
    acc = collections.OrderedDict()
    for d in disp:
        a = acc.setdefault(d["k"], [0, collections.Counter()])
        a[0] += 1
        for n in names:
            a[1][n] += d.get(n, 0.0)
    with open(dst, "w") as f:
        f.write("kernel,grid,workgroup,launches_per_step," + ",".join("avg_" + n for n in names) + "\n")
        for (k, g, w), (n, c) in acc.items():
            f.write(f"\"{k}\",{g},{w},{n / n_steps:.2f}," + ",".join(f"{c[x] / n:.1f}" for x in names) + "\n")
        f.write("\"TOTAL per diffusion step\",,,%.1f," % (len(disp) / n_steps) + ",".join(f"{sum(c[x] for _, c in acc.values()) / n_steps:.1f}" for x in names) + "\n")
    print(f"reduce: {len(disp)} dispatches, counters {names}")


def cmd_merge(d, dst):
    tab = collections.OrderedDict()
    for fn in sorted(os.listdir(d)):
        if not fn.endswith(".csv"):
            continue
        for r in csv.DictReader(open(os.path.join(d, fn))):
            key = (r["kernel"], r.get("grid", ""))
            row = tab.setdefault(key, {"kernel": r["kernel"], "grid": r.get("grid", "")})
            for k, v in r.items():
                if k not in ("kernel", "grid", "workgroup") and v not in ("", None):
                    row[k] = float(v)
            row["workgroup"] = r.get("workgroup", "")
    json.dump(list(tab.values()), open(dst, "w"), indent=1)
    print(f"merged {len(tab)} rows -> {dst}")


if __name__ == "__main__":
    if sys.argv[1] == "merge":
        cmd_merge(sys.argv[2], sys.argv[3])
    else:
        (cmd_trace if sys.argv[1] == "trace" else cmd_reduce)(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
