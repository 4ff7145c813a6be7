#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/pp1; mkdir -p $O
timeout 600 python tools/check_tiles.py 2>&1 | tee $O/check_tiles.txt | tail -5
FDM_TILE_EXTRA=1 timeout 600 python tools/bench_gemm_tiles.py bf16 6400 1024 2048 6400 2048 1024 6400 3072 1024 6400 1024 1024 8192 1024 2048 4096 4096 4096 2>&1 | grep -E "tile (auto|128x128|256x128|128x64_s3|tile10)" | tee $O/gemm_tiles_bf16.txt
