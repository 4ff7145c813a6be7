#!/bin/bash
# round 3: split attention at head_dim 256 (BIWI contract mode), the VQ return tuple, cfg4 before / after
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_hubert_vq_gpu.py -m gpu -x -q -k "split_attention or attention or vq" > $O/pytest_ops.log 2>&1; echo "pytest ops rc=$?"; tail -3 $O/pytest_ops.log
timeout 1500 python -m pytest tests/test_denoiser_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "biwi or cfg4" > $O/pytest_biwi.log 2>&1; echo "pytest biwi rc=$?"; tail -3 $O/pytest_biwi.log
for v in 0 1; do
  FDM_ATTN_HD256_F32=$v timeout 600 python bench.py --config cfg4 --dtype f16x3 --headline-only --no-cpu-baseline --steps 5 > $O/cfg4_f16x3_f32attn$v.json 2> $O/cfg4_f16x3_f32attn$v.err
  python3 -c "import json; d=json.load(open('$O/cfg4_f16x3_f32attn$v.json')); r=d['roofline']; print('cfg4 f16x3 FDM_ATTN_HD256_F32=$v', d['value'], 'frames/s', r['avg_launch_ms'], 'ms/step launches', d['kernel_launches_per_diffusion_step'], d.get('gemm_tiles'))" || tail -3 $O/cfg4_f16x3_f32attn$v.err
done
