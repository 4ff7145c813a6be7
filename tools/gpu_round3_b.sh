#!/bin/bash
# round 3: row-count sweep with the ping-pong tile in the tuner's candidate list, the tile test, counters at B = 32
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
timeout 900 python -m pytest tests/test_denoiser_gpu.py -m gpu -x -q -k "tile_choice or condition" > $O/pytest_tiles.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_tiles.log
ROWS_SWEEP_NO_GEMM=1 ROWS_SWEEP_B="4 8 16 32 40" bash tools/rows_sweep.sh 2>&1 | grep -v "^gemm" 
mkdir -p gpurun_out/r3b/rows_sweep; cp gpurun_out/rows_sweep/* gpurun_out/r3b/rows_sweep/ 2>/dev/null
bash tools/pmc_collect.sh cfg2_B32_bf16 --config cfg2 --dtype bf16 --batch 32 --headline-only > $O/pmc.log 2>&1; tail -3 $O/pmc.log
