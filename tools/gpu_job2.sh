#!/bin/bash
# round-2 job 2: split-operand modes: op tests, denoiser parity tests, bench of every mode
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job2; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm or split" > $O/ops.log 2>&1
tail -5 $O/ops.log
timeout 1500 python -m pytest tests/test_denoiser_gpu.py -x -q -m gpu -s > $O/den.log 2>&1
tail -15 $O/den.log
for dt in f16x3 bf16x3; do
  timeout 600 python bench.py --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$dt.json 2> $O/bench_$dt.err
  cat $O/bench_$dt.json | cut -c1-400
done
