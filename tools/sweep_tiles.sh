#!/bin/bash
# A/B of the GEMM tile thresholds on whole-step time (run on the GPU box)
for c in cfg5 cfg3 cfg2; do
  for t in "512 700" "512 256" "512 200" "512 100" "128 700" "128 256" "64 128" "256 256"; do
    set -- $t
    r=$(FDM_GEMM_T128=$1 FDM_GEMM_T128X64=$2 python bench.py --config $c --dtype bf16 --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$c T128=$1 T128X64=$2 -> $r"
  done
done
