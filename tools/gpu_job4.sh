#!/bin/bash
# round-2 job 4: full GPU suite on the C++ plan layer + bench of every mode
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job4; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu -s > $O/gpu_tests.log 2>&1
tail -25 $O/gpu_tests.log
for dt in bf16 f16x3 f32; do
  timeout 600 python bench.py --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$dt.json 2> $O/bench_$dt.err
  cut -c1-330 $O/bench_$dt.json; tail -2 $O/bench_$dt.err
done
