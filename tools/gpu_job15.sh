#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job15; mkdir -p $O
timeout 1800 python -m pytest tests/test_denoiser_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do
timeout 600 python bench.py --config cfg3 --steps 3 --warmup 1 --no-cpu-baseline > $O/cfg3_$i.json 2>/dev/null
python3 -c "import json; d=json.load(open('$O/cfg3_$i.json')); print('cfg3', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['kernel_launches_per_diffusion_step'])"
done
