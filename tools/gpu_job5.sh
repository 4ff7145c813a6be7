#!/bin/bash
# round-2 job 5: cooperative L2 warm-up A/B, graph steps per launch A/B
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job5; mkdir -p $O
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $BARGS > $O/$name.json 2> $O/$name.err
  python3 -c "import json,sys; d=json.load(open('$O/$name.json')); print('$name', d['dtype'], d['value'], 'frames/s', d['roofline']['avg_launch_ms'], 'ms/step')" 2>/dev/null || tail -3 $O/$name.err
}
BARGS="--dtype bf16"
run bf16_pf0 FDM_X=0
run bf16_pf1 FDM_GEMM_PREFETCH=1
run bf16_pf2 FDM_GEMM_PREFETCH=2
run bf16_pf3 FDM_GEMM_PREFETCH=3
run bf16_pf0b FDM_X=0
run bf16_gs1 FDM_GRAPH_STEPS=1
run bf16_gs50 FDM_GRAPH_STEPS=50
BARGS="--dtype f16x3"
run f16x3_pf0 FDM_X=0
run f16x3_pf1 FDM_GEMM_PREFETCH=1
run f16x3_pf3 FDM_GEMM_PREFETCH=3
BARGS="--dtype bf16 --config cfg3"
run cfg3_pf0 FDM_X=0
run cfg3_pf3 FDM_GEMM_PREFETCH=3
BARGS="--dtype bf16 --config cfg5"
run cfg5_pf0 FDM_X=0
run cfg5_pf3 FDM_GEMM_PREFETCH=3
