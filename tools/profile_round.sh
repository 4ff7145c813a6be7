#!/bin/bash
# The round's committed profiles: counter passes (tools/pmc_collect.sh) with the tiles pinned, for the benched configuration
# in every arithmetic mode and for cfg3 / cfg5; results are copied to profiles/r2_pmc_<config>_<dtype>/ by the caller
# (gpurun_out/ is scratch).  Also one plain `rocprofv3 --kernel-trace --stats` of the default bench command.
cd "$GRAFT_REPO_ROOT"
export FDM_TUNE=0
FDM_TILE_OVERRIDE="qkv=3" bash tools/pmc_collect.sh cfg2_bf16 --dtype bf16 2>&1 | tail -3
FDM_TILE_OVERRIDE="qkv=3,ffn1=8" bash tools/pmc_collect.sh cfg2_f16x3 --dtype f16x3 2>&1 | tail -3
FDM_TILE_OVERRIDE="" bash tools/pmc_collect.sh cfg2_f32 --dtype f32 2>&1 | tail -3
FDM_TILE_OVERRIDE="enc=9,qkv=7,out=9,ffn1=6,ffn2=2,dec=9" bash tools/pmc_collect.sh cfg3_bf16 --dtype bf16 --config cfg3 2>&1 | tail -3
FDM_TILE_OVERRIDE="enc=6,qkv=7,out=2,ffn1=3,ffn2=2,dec=2" bash tools/pmc_collect.sh cfg5_bf16 --dtype bf16 --config cfg5 2>&1 | tail -3
unset FDM_TUNE FDM_TILE_OVERRIDE
cd /tmp; export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/stats_default
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/stats_default -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/stats_default.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/stats_default -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/stats_default_kernel_stats.csv \;
find $GRAFT_REPO_ROOT/gpurun_out/stats_default -name "*kernel_trace.csv" -delete
tail -2 $GRAFT_REPO_ROOT/gpurun_out/stats_default.log
