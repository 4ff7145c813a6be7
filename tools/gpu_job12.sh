#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job12; mkdir -p $O
timeout 1800 python -m pytest tests/test_ops_gpu.py tests/test_denoiser_gpu.py tests/test_abi_c_gpu.py -x -q -m gpu -s > $O/tests.log 2>&1
tail -12 $O/tests.log
for dt in f16x3 bf16 bf16x3; do
timeout 600 python bench.py --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$dt.json 2> $O/bench_$dt.err
python3 -c "import json; d=json.load(open('$O/bench_$dt.json')); print(d['dtype'], d['value'], 'frames/s', d['roofline']['avg_launch_ms'], 'ms/step', d['roofline']['frac'], d.get('gemm_tiles'))" || tail -3 $O/bench_$dt.err
done
