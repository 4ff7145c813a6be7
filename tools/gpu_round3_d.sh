#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O
timeout 600 python tools/measure_bf16_bars.py 2>&1 | grep -v amdgpu.ids | tee $O/bf16_bars.txt
timeout 1800 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -s -k "cfg5" 2>&1 | grep -E "cfg5|passed|failed|Error" | tee $O/cfg5.txt
