#!/bin/bash
# Round 3: where do the step's kernels stand when the row count is not the limiter?  cfg2's shape at B clips per GPU
# (M = 200 B rows) in bf16 and f16x3 with the tuner on, plus isolated per-tile GEMM timings at M = 6400.
# Output: gpurun_out/rows_sweep/{B<clips>_<dtype>.json, gemm_tiles_<dtype>.txt}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/rows_sweep; mkdir -p $O
for dt in bf16 f16x3; do
  for B in ${ROWS_SWEEP_B:-4 8 16 32}; do
    FDM_TUNE_VERBOSE=1 timeout 600 python bench.py --config cfg2 --dtype $dt --batch $B --steps 2 --warmup 1 --no-cpu-baseline --headline-only > $O/B${B}_$dt.json 2> $O/B${B}_$dt.err
    python3 -c "import json; d=json.load(open('$O/B${B}_$dt.json')); r=d['roofline']; print('B=$B', d['dtype'], d['value'], 'frames/s', r['avg_launch_ms'], 'ms/step', r['achieved'], 'TF', r['frac'], d.get('gemm_tiles'))" || tail -3 $O/B${B}_$dt.err
  done
done
[ -n "$ROWS_SWEEP_NO_GEMM" ] && exit 0
for dt in bf16 f16x3; do
  FDM_GEMM_TILES_JSON=$O/gemm_tiles_$dt.json timeout 600 python tools/bench_gemm_tiles.py $dt 6400 1024 2048 6400 2048 1024 6400 3072 1024 6400 1024 1024 800 1024 1024 2>&1 | tee $O/gemm_tiles_$dt.txt
done
