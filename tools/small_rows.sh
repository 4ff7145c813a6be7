#!/bin/bash
# Round 5, review item 3: the bs = 1 regime the reference's callers issue (100-250 rows: cfg1, shipped_mead, shipped_biwi), same box,
# alternating: plain step program | norm3 folded (fuse_ln3) | single-clip setting (K slices 2 / 4 on out-proj / FFN2, the default
# of these configs) | both.   bash tools/small_rows.sh [outdir]      -> <outdir>/table.md + one JSON line per run
out=${1:-gpurun_out/r5_small_rows}
mkdir -p $out
for rep in 1 2; do
for c in cfg1 shipped_mead shipped_biwi; do for m in bf16 f16x3; do
  for v in plain fuse single single_fuse; do
    case $v in
      plain) o="--plan-set ksplit.out=1 --plan-set ksplit.ffn2=1";;
      fuse) o="--plan-set ksplit.out=1 --plan-set ksplit.ffn2=1 --plan-set fuse_ln3=1";;
      single) o="";;
      single_fuse) o="--plan-set fuse_ln3=1";;
    esac
    timeout 300 python bench.py --config $c --dtype $m --headline-only --no-cpu-baseline --steps 5 $o > $out/${c}_${m}_${v}_$rep.json 2> $out/${c}_${m}_${v}_$rep.err
  done
done; done
done
python - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
rows = []
for c in ("cfg1", "shipped_mead", "shipped_biwi"):
    for m in ("bf16", "f16x3"):
        r = [c, m]
        base = None
        for v in ("plain", "fuse", "single", "single_fuse"):
            vals = []
            for f in sorted(glob.glob(f"{out}/{c}_{m}_{v}_*.json")):
                try:
                    d = json.load(open(f)); vals.append((d["value"], d["roofline"]["avg_launch_ms"], d["kernel_launches_per_diffusion_step"]))
                except Exception:
                    pass
            if not vals:
                r.append("-"); continue
            best = max(vals)
            if v == "plain": base = best[0]
            r.append(f"{best[0]:.0f} \\| {best[1]:.4f} ms \\| {best[2]} launches" + (f" \\| {100 * (best[0] / base - 1):+.1f} %" if base and v != "plain" else ""))
        rows.append(r)
with open(os.path.join(out, "table.md"), "w") as f:
    f.write("| config | mode | plain (frames/s \\| step \\| launches) | fuse_ln3 | single-clip setting (ksplit 2 / 4) | both |\n|---|---|---|---|---|---|\n")
    for r in rows:
        f.write("| " + " | ".join(r) + " |\n")
print(open(os.path.join(out, "table.md")).read())
PY
