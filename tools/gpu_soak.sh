#!/bin/bash
# Determinism soak + tile-independence check on the GPU box:  gpurun -- 'bash tools/gpu_soak.sh'
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1500 python tools/soak_determinism.py 2>&1 | grep -v amdgpu.ids | tail -10
timeout 300 python tools/check_tiles.py 2>&1 | tail -3
