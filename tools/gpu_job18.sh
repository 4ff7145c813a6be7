#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job18; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "row_layernorm" 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_denoiser_gpu.py -x -q -m gpu 2>&1 | tail -15
for dt in bf16 f16x3 f32; do
  timeout 600 python bench.py --dtype $dt --steps 3 --warmup 1 2>$O/bench_$dt.err | tee $O/bench_lnx_$dt.json | cut -c1-400
  FDM_FUSE_LNX=0 timeout 600 python bench.py --dtype $dt --steps 3 --warmup 1 2>>$O/bench_$dt.err | tee $O/bench_nolnx_$dt.json | cut -c1-400
done
