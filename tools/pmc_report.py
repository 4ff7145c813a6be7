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


_DEMANGLED = {}


def short(name):
    if name.startswith("_Z"):
        if name not in _DEMANGLED:
            import shutil
            import subprocess
            tool = shutil.which("llvm-cxxfilt") or shutil.which("c++filt") or "c++filt"
            try:      # (GNU c++filt does not know DF16b = __bf16: spell it as a vendor type)
                _DEMANGLED[name] = subprocess.run([tool, name.replace("DF16b", "u6__bf16")], capture_output=True, text=True, timeout=10).stdout.strip() or name
            except Exception:
                _DEMANGLED[name] = name
        name = _DEMANGLED[name]
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


def cmd_derive(d, dst, meta):
    """One row per kernel of the step graph from trace.csv + the counter passes of pmc_collect.sh, with derived shares:
      mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32 shader engines
                    (SQ_VALU_MFMA_BUSY_CYCLES = 16 x number of 16x16x32 MFMAs: checked against SQ_INSTS_MFMA)
      wait / stall / issue = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint, sum ~ 1)
      lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;  l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS)
      l1_l2_latency = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ (cycles)
      fetch / write MB: FETCH_SIZE x 2 (gfx950 tallies 16-B-per-lane streams at 1/2: MI355X_MICROARCH.md, HBM) and WRITE_SIZE, KB -> MB"""
    tabs = {}
    for fn in ("trace", "sq1", "sq2", "fetch", "write", "tcp"):
        path = os.path.join(d, fn + ".csv")
        if os.path.exists(path):
            tabs[fn] = list(csv.DictReader(open(path)))
    rows = []
    tr = tabs["trace"]
    def find(tab, i, kernel):
        # passes list kernels in first-appearance order of the same step program: match by position, check the name
        r = tabs[tab][i] if tab in tabs and i < len(tabs[tab]) else None
        return r if r is not None and short(r["kernel"]) == short(kernel) else None
    tot = {"us": 0.0, "fetch": 0.0, "write": 0.0, "mfma": 0.0}
    for i, r in enumerate(tr):
        if r["kernel"].startswith("TOTAL") or r["kernel"].startswith("wall"):
            continue
        o = {"kernel": short(r["kernel"]), "grid": r["grid"], "launches_per_step": float(r["launches_per_step"]), "avg_us": float(r["avg_us"])}
        s1, s2, fe, wr, tc = (find(t, i, r["kernel"]) for t in ("sq1", "sq2", "fetch", "write", "tcp"))
        if s1:
            cyc = float(s1["avg_SQ_BUSY_CYCLES"]) / 32.0
            wc = float(s1["avg_SQ_WAVE_CYCLES"])
            o.update(kernel_cycles=round(cyc), clock_ghz=round(cyc / (o["avg_us"] * 1e3), 2),
                     mfma_busy=round(float(s1["avg_SQ_VALU_MFMA_BUSY_CYCLES"]) / (1024.0 * cyc), 4),
                     wait=round(float(s1["avg_SQ_WAIT_ANY"]) / wc, 3), stall=round(float(s1["avg_SQ_WAIT_INST_ANY"]) / wc, 3),
                     issue=round(float(s1["avg_SQ_ACTIVE_INST_ANY"]) / wc, 3), lds_issue_stall=round(float(s1["avg_SQ_WAIT_INST_LDS"]) / wc, 3),
                     waves=float(s1["avg_SQ_WAVES"]))
        if s2:
            ia = float(s2["avg_SQ_LDS_IDX_ACTIVE"])
            o.update(mfma_insts=float(s2["avg_SQ_INSTS_MFMA"]), lds_conflict=round(float(s2["avg_SQ_LDS_BANK_CONFLICT"]) / ia, 3) if ia else 0.0)
        if fe:
            o["fetch_mb"] = round(float(fe["avg_FETCH_SIZE"]) * 2 / 1024.0, 2)
        if wr:
            o["write_mb"] = round(float(wr["avg_WRITE_SIZE"]) / 1024.0, 2)
        if fe and wr:
            h, m = float(fe["avg_TCC_HIT_sum"]), float(wr["avg_TCC_MISS_sum"])
            o["l2_hit"] = round(h / (h + m), 3) if h + m else None
        if tc and float(tc["avg_TCP_TCC_READ_REQ_sum"]):
            o["l1_l2_latency_cycles"] = round(float(tc["avg_TCP_TCC_READ_REQ_LATENCY_sum"]) / float(tc["avg_TCP_TCC_READ_REQ_sum"]))
        n = o["launches_per_step"]
        tot["us"] += n * o["avg_us"]
        tot["fetch"] += n * o.get("fetch_mb", 0.0)
        tot["write"] += n * o.get("write_mb", 0.0)
        tot["mfma"] += n * o.get("mfma_busy", 0.0) * o["avg_us"]
        rows.append(o)
    summary = {"kernel_us_per_step": round(tot["us"], 1), "fetch_mb_per_step": round(tot["fetch"], 1), "write_mb_per_step": round(tot["write"], 1),
               "traffic_bytes_per_step": round((tot["fetch"] + tot["write"]) * 1024 * 1024),
               "mfma_busy_time_weighted": round(tot["mfma"] / tot["us"], 4) if tot["us"] else None,
               "launches_per_step": round(sum(r["launches_per_step"] for r in rows), 1)}
    summary.update(meta)
    json.dump({"summary": summary, "kernels": rows}, open(dst, "w"), indent=1)
    md = dst.replace(".json", ".md")
    cols = ["kernel", "grid", "launches_per_step", "avg_us", "mfma_busy", "wait", "stall", "issue", "lds_conflict", "l2_hit", "l1_l2_latency_cycles", "fetch_mb", "write_mb", "clock_ghz"]
    with open(md, "w") as f:
        f.write("| " + " | ".join(cols) + " |\n|" + "---|" * len(cols) + "\n")
        for r in rows:
            f.write("| " + " | ".join(str(r.get(c, "")) for c in cols) + " |\n")
        f.write("\nsummary: " + json.dumps(summary) + "\n")
    print(json.dumps(summary))


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
    if sys.argv[1] == "derive":
        cmd_derive(sys.argv[2], sys.argv[3], json.loads(sys.argv[4]) if len(sys.argv) > 4 else {})
    elif sys.argv[1] == "merge":
        cmd_merge(sys.argv[2], sys.argv[3])
    else:
        (cmd_trace if sys.argv[1] == "trace" else cmd_reduce)(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
