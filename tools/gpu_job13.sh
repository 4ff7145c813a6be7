#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job13; mkdir -p $O
timeout 600 python tools/check_tiles.py 2>&1 | tail -8
timeout 900 python tools/gemm_insitu.py bf16 800 0 1 14 17 2 15 3 16 8 7 2>&1 | grep -v amdgpu.ids
timeout 900 python tools/gemm_insitu.py bf16 1992 0 1 14 2 15 3 16 7 2>&1 | grep -v amdgpu.ids
timeout 900 python tools/gemm_insitu.py f32 800 0 1 14 2 15 2>&1 | grep -v amdgpu.ids
