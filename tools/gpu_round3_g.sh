#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_denoiser_gpu.py tests/test_abi_c_gpu.py -m gpu -x -q -k "pingpong or tile_choice or two_rank" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for B in 32 40; do
  FDM_TUNE_VERBOSE=1 timeout 600 python bench.py --config cfg2 --dtype bf16 --batch $B --steps 2 --warmup 1 --no-cpu-baseline --headline-only > $O/B${B}_bf16.json 2> $O/B${B}_bf16.err
  python3 -c "import json; d=json.load(open('$O/B${B}_bf16.json')); r=d['roofline']; print('B=$B', d['value'], 'frames/s', r['avg_launch_ms'], 'ms/step', r['frac'], d.get('gemm_tiles'))" || tail -3 $O/B${B}_bf16.err
done
