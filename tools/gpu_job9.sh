#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job9; mkdir -p $O
run() { local name=$1; shift
  env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $BARGS > $O/$name.json 2> $O/$name.err
  python3 -c "import json,sys; d=json.load(open('$O/$name.json')); print('$name', d['dtype'], d['value'], 'frames/s', d['roofline']['avg_launch_ms'], 'ms/step', d.get('gemm_tiles'))" 2>/dev/null || tail -3 $O/$name.err
}
BARGS="--dtype bf16"
run bf16_new FDM_X=0
run bf16_oldln FDM_LN_ROW_MAX=0
run bf16_new2 FDM_X=0
BARGS="--dtype f16x3"
run f16x3_new FDM_X=0
run f16x3_oldln FDM_LN_ROW_MAX=0
BARGS="--dtype bf16 --config cfg3"
run cfg3_new FDM_X=0
run cfg3_oldln FDM_LN_ROW_MAX=0
BARGS="--dtype bf16 --config cfg5"
run cfg5_new FDM_X=0
run cfg5_oldln FDM_LN_ROW_MAX=0
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_denoiser_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
