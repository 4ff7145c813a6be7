#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job14; mkdir -p $O
run() { local name=$1; shift
  env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $BARGS > $O/$name.json 2> $O/$name.err
  python3 -c "import json,sys; d=json.load(open('$O/$name.json')); print('$name', d['dtype'], d['value'], 'frames/s', d['roofline']['avg_launch_ms'], 'ms/step', d.get('gemm_tiles'))" 2>/dev/null || tail -3 $O/$name.err
}
BARGS="--dtype bf16"
run bf16_a FDM_X=0
run bf16_qs2 FDM_ATTN_QS2=128
run bf16_b FDM_X=0
run bf16_qs2b FDM_ATTN_QS2=128
BARGS="--dtype bf16 --config cfg3"
run cfg3_a FDM_X=0
run cfg3_qs2 FDM_ATTN_QS2=128
