#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job16; mkdir -p $O
FDM_FUSE_LN3=1 timeout 1800 python -m pytest tests/test_denoiser_gpu.py -x -q -m gpu -s -k "fp32 or f16x3 or cfg2 or golden or tile" > $O/tests.log 2>&1; grep -E "max-abs|passed|failed|Error" $O/tests.log | tail -6
for e in 0 1 0 1; do
FDM_FUSE_LN3=$e timeout 600 python bench.py --dtype f16x3 --steps 3 --warmup 1 --no-cpu-baseline > $O/f16x3_$e.json 2>/dev/null
python3 -c "import json; d=json.load(open('$O/f16x3_$e.json')); print('f16x3 fuse=$e', d['value'], d['roofline']['avg_launch_ms'], d['kernel_launches_per_diffusion_step'], d.get('gemm_tiles'))"
done
