#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "row_layernorm" 2>&1 | tail -5
bash tools/gpu_job19.sh 2>&1 | grep "true>\|TOTAL\|trace:"
