#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/mid; mkdir -p $O
timeout 300 python tools/check_tiles.py 2>&1 | tail -2
FDM_TILE_EXTRA=3 timeout 900 python tools/bench_gemm_tiles.py bf16 1992 1024 1024 1992 1024 2048 1992 2048 1024 1992 3072 1024 2400 512 512 2400 1536 512 2400 1024 512 2400 512 1024 3200 1024 1024 3200 1024 2048 3200 3072 1024 2>&1 | grep "gemm bf16" | tee $O/gemm_tiles_bf16.txt
