#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job11; mkdir -p $O
timeout 2400 python -m pytest tests/test_hubert_vq_gpu.py tests/test_pipeline_gpu.py tests/test_metrics_gpu.py tests/test_configs_gpu.py -x -q -m gpu -s > $O/tests.log 2>&1
tail -25 $O/tests.log
timeout 600 python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; cut -c1-300 $O/bench_cfg5.json; tail -2 $O/bench_cfg5.err
timeout 600 python tools/bench_e2e.py bf16 > $O/e2e.txt 2>&1; tail -8 $O/e2e.txt
