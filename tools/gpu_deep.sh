#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/deep; mkdir -p $O
FDM_TILE_EXTRA=5 timeout 900 python tools/bench_gemm_tiles.py bf16 800 1024 1024 800 1024 2048 800 2048 1024 800 3072 1024 1992 1024 1024 1992 1024 2048 2>&1 | grep -E "tile (auto|64x64|128x64|128x128|64x64_s3|128x64_s3|tile1[0-4]) " | tee $O/gemm_tiles_bf16.txt
