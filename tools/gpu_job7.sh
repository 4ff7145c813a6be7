#!/bin/bash
# round-2 job 7: L2 warm-up with plain (allocating) loads; remaining new tests
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job7; mkdir -p $O
run() { local name=$1; shift
  env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $BARGS > $O/$name.json 2> $O/$name.err
  python3 -c "import json,sys; d=json.load(open('$O/$name.json')); print('$name', d['dtype'], d['value'], 'frames/s', d['roofline']['avg_launch_ms'], 'ms/step')" 2>/dev/null || tail -3 $O/$name.err
}
BARGS="--dtype bf16"
run bf16_pf0 FDM_X=0
run bf16_pf5 FDM_GEMM_PREFETCH=5
run bf16_pf6 FDM_GEMM_PREFETCH=6
run bf16_pf7 FDM_GEMM_PREFETCH=7
run bf16_pf0b FDM_X=0
BARGS="--dtype f16x3"
run f16x3_pf0 FDM_X=0
run f16x3_pf5 FDM_GEMM_PREFETCH=5
run f16x3_pf7 FDM_GEMM_PREFETCH=7
timeout 1800 python -m pytest tests/test_abi_c_gpu.py tests/test_configs_gpu.py -x -q -m gpu -s > $O/tests.log 2>&1
tail -15 $O/tests.log
