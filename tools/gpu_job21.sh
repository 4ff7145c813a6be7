#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job21; mkdir -p $O
for cfg in cfg3 cfg5; do for dt in bf16 f16x3; do for lnx in 1 0; do
  FDM_FUSE_LNX=$lnx timeout 900 python bench.py --config $cfg --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline 2>>$O/err.log > $O/b_${cfg}_${dt}_$lnx.json
  python - <<PY
import json
j=json.loads(open("$O/b_${cfg}_${dt}_$lnx.json").read())
print("$cfg $dt lnx=$lnx", j["value"], j["ms_per_step"], j.get("kernel_launches_per_diffusion_step"), j.get("gemm_tiles"))
PY
done; done; done
