#!/bin/bash
# round-2 job 3: counter list, f16x3 kernel trace, AMD_DIRECT_DISPATCH A/B
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job3; mkdir -p $O
(cd /tmp && timeout 120 rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters.txt 2>&1)
AMD_DIRECT_DISPATCH=0 timeout 600 python bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_bf16_nodirect.json 2> $O/bench_bf16_nodirect.err
timeout 600 python bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_f16x3 -- python3 $GRAFT_REPO_ROOT/bench.py --dtype f16x3 --steps 1 --warmup 0 --no-cpu-baseline --profile-steps 30 > $GRAFT_REPO_ROOT/$O/prof_f16x3.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r2_job3/prof_f16x3/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'fdm' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
tail = rows[-30*59:]
acc = collections.OrderedDict()
for r in tail:
    k = (r['Kernel_Name'][:110], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''))
    a = acc.setdefault(k, [0, 0]); a[0] += 1; a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
with open('gpurun_out/r2_job3/f16x3_step_kernels.txt', 'w') as o:
    tot = 0
    for (k, g), (n, ns) in acc.items():
        o.write(f"{k} grid={g} per_step={n/30:.1f} avg_us={ns/n/1e3:.2f} step_us={ns/30/1e3:.1f}\n"); tot += ns
    span = (int(tail[-1]['End_Timestamp']) - int(tail[0]['Start_Timestamp'])) / 30 / 1e3
    o.write(f"kernel time per step {tot/30/1e3:.1f} us; wall span per step {span:.1f} us\n")
print(open('gpurun_out/r2_job3/f16x3_step_kernels.txt').read())
PY
find $O -name "*kernel_trace.csv" -delete
