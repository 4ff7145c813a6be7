#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job18; mkdir -p $O
timeout 1200 python -m pytest tests/test_denoiser_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "biwi" 2>&1 | tail -6
for dt in bf16 f16x3 f32; do
  timeout 600 python bench.py --config cfg4 --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline 2>>$O/err.log | cut -c1-330
done
