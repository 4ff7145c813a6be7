#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O
bash tools/bench_all.sh 2>&1 | grep -v amdgpu.ids
for f in 0 1; do
  FDM_FUSE_LN3=$f timeout 600 python bench.py --config cfg2 --dtype bf16 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --headline-only > $O/B32_bf16_fuse$f.json 2> $O/B32_bf16_fuse$f.err
  python3 -c "import json; d=json.load(open('$O/B32_bf16_fuse$f.json')); r=d['roofline']; print('B=32 bf16 FUSE_LN3=$f', d['value'], 'frames/s', r['avg_launch_ms'], 'ms/step', r['frac'], 'launches', d['kernel_launches_per_diffusion_step'], d.get('gemm_tiles'))" || tail -3 $O/B32_bf16_fuse$f.err
done
