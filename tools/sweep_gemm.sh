#!/bin/bash
# per-GEMM A/B of tile variants (run on the GPU box): bench_ops prints us per launch for the step's GEMM shapes
for M in 800 1992; do
  for v in 1 3 4 6; do
    echo "== M=$M variant=$v"
    FDM_GEMM_VARIANT=$v python tools/bench_ops.py bf16 $M 2>&1 | grep "^gemm"
  done
done
