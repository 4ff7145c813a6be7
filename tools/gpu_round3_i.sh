#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r3i; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_hubert_vq_gpu.py -m gpu -x -q -k "attention or hubert_vs_golden or vq_quant_decode" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for cd in "cfg4 f16x3" "cfg4 bf16" "cfg2 bf16" "cfg2 f16x3" "cfg5 bf16"; do
  set -- $cd
  timeout 600 python bench.py --config $1 --dtype $2 --headline-only --no-cpu-baseline --steps 3 > $O/$1_$2.json 2> $O/$1_$2.err
  python3 -c "import json; d=json.load(open('$O/$1_$2.json')); r=d['roofline']; print('$1 $2', d['value'], 'frames/s', r['avg_launch_ms'], 'ms/step', d.get('gemm_tiles'))" || tail -3 $O/$1_$2.err
done
