#!/bin/bash
# final-build counter profiles of the headline configuration in both driver-timed modes (tiles pinned: reproducible)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
FDM_TUNE=0 FDM_TILE_OVERRIDE="qkv=3" bash tools/pmc_collect.sh cfg2_bf16 --config cfg2 --dtype bf16 --headline-only > gpurun_out/pmc_r3_cfg2_bf16.log 2>&1; tail -1 gpurun_out/pmc_r3_cfg2_bf16.log
FDM_TUNE=0 FDM_TILE_OVERRIDE="qkv=3,ffn1=8" bash tools/pmc_collect.sh cfg2_f16x3 --config cfg2 --dtype f16x3 --headline-only > gpurun_out/pmc_r3_cfg2_f16x3.log 2>&1; tail -1 gpurun_out/pmc_r3_cfg2_f16x3.log
FDM_TUNE=0 FDM_TILE_OVERRIDE="qkv=3" bash tools/pmc_collect.sh cfg1x8_bf16 --config cfg1x8 --dtype bf16 --headline-only > gpurun_out/pmc_r3_cfg1x8_bf16.log 2>&1; tail -1 gpurun_out/pmc_r3_cfg1x8_bf16.log
for t in cfg2_bf16 cfg2_f16x3 cfg1x8_bf16; do echo "== $t"; cut -c1-230 gpurun_out/pmc_$t/summary.md; done
