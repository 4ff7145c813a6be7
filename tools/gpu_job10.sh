#!/bin/bash
# round-2 job 10: counter profiles with pinned tiles: cfg2 bf16 / f16x3, cfg3 bf16, cfg5 bf16
cd "$GRAFT_REPO_ROOT"
export FDM_TUNE=0
FDM_TILE_OVERRIDE="qkv=7,qkv_ln=7,enc=1,dec_ln=1" bash tools/pmc_collect.sh cfg2_bf16 --dtype bf16 2>&1 | tail -4
FDM_TILE_OVERRIDE="qkv=3,ffn1=8" bash tools/pmc_collect.sh cfg2_f16x3 --dtype f16x3 2>&1 | tail -4
FDM_TILE_OVERRIDE="qkv=7,qkv_ln=7,out=9,out_ln=9,ffn1=8,ffn2_stat=9" bash tools/pmc_collect.sh cfg3_bf16 --dtype bf16 --config cfg3 2>&1 | tail -4
FDM_TILE_OVERRIDE="enc=6,qkv=7,qkv_ln=7,out=2,out_ln=2,ffn1=7,ffn2_stat=2,dec_ln=2" bash tools/pmc_collect.sh cfg5_bf16 --dtype bf16 --config cfg5 2>&1 | tail -4
