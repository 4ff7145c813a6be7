#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
bash tools/gpu_full_suite.sh
S=$SECONDS; python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench wall $((SECONDS-S)) s"
python3 -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline'], d.get('gemm_tiles')); c=d['contract_mode']; print(c['value'], c['roofline']['frac'], c.get('parity'), c.get('gemm_tiles')); print(d['parity'])"
FDM_TUNE=0 python bench.py --headline-only --no-cpu-baseline | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('untuned (heuristic) cfg2 bf16', d['value'], d['roofline']['avg_launch_ms'])"
