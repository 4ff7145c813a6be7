#!/bin/bash
# Every configuration in every arithmetic mode it supports, one JSON line each -> gpurun_out/bench_all/<config>_<dtype>.json
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/bench_all; mkdir -p $O
for c in cfg2 cfg1 cfg3 cfg4 cfg5; do
  for dt in bf16 f16 f16x3 f32; do
    steps=3; if [ $dt = f32 ]; then steps=2; fi
    extra="--no-cpu-baseline"; if [ $c = cfg2 ] && [ $dt = bf16 ]; then extra=""; fi
    timeout 900 python bench.py --config $c --dtype $dt --steps $steps --warmup 1 --headline-only $extra > $O/${c}_$dt.json 2> $O/${c}_$dt.err
    python3 -c "import json; d=json.load(open('$O/${c}_$dt.json')); r=d['roofline']; print('$c', d['dtype'], d['value'], 'frames/s', r['avg_launch_ms'], 'ms/step', r['achieved'], 'TF', r['frac'], 'launches', d['kernel_launches_per_diffusion_step'], 'graph launches', d['host_graph_launches_per_sample'], d.get('gemm_tiles'))" || tail -3 $O/${c}_$dt.err
  done
done
