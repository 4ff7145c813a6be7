#!/bin/bash
# The round's committed evidence in one GPU job (round 5): counter passes of the benched configuration (tools/pmc_collect.sh, heuristic
# tiles pinned with FDM_TUNE=0), the driver's default bench line, the reference callers' per-clip workloads (bench.py --config
# shipped_*), the once-per-clip stages under rocprofv3 (tools/profile_encoders.sh), and one plain `rocprofv3 --kernel-trace --stats` of
# the default bench command.  Results land in gpurun_out/ (scratch); the caller copies what is to be judged into profiles/r5_*.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5_bench; mkdir -p $O
FDM_TUNE=0 FDM_TILE_OVERRIDE="" bash tools/pmc_collect.sh r5_cfg2_bf16 --dtype bf16 --headline-only 2>&1 | tail -2
FDM_TUNE=0 FDM_TILE_OVERRIDE="" bash tools/pmc_collect.sh r5_cfg2_f16x3 --dtype f16x3 --headline-only 2>&1 | tail -2
cd "$GRAFT_REPO_ROOT"
# the committed counter summaries must exist BEFORE the bench lines that quote them are taken
for t in cfg2_bf16 cfg2_f16x3; do mkdir -p profiles/r5_pmc_$t; cp gpurun_out/pmc_r5_$t/summary.json gpurun_out/pmc_r5_$t/summary.md profiles/r5_pmc_$t/ 2>/dev/null; done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
for c in shipped_vocaset shipped_mead shipped_biwi cfg1x8 cfg1; do
  timeout 900 python bench.py --config $c --dtype bf16 --headline-only --steps 3 --warmup 1 > $O/${c}_bf16.json 2> $O/${c}_bf16.err
done
for c in shipped_mead shipped_biwi cfg1; do
  timeout 900 python bench.py --config $c --dtype f16x3 --headline-only --steps 3 --warmup 1 --no-cpu-baseline > $O/${c}_f16x3.json 2> $O/${c}_f16x3.err
done
timeout 900 python bench.py --config shipped_vocaset --dtype f16x3 --headline-only --steps 3 --warmup 1 --no-cpu-baseline > $O/shipped_vocaset_f16x3.json 2> $O/shipped_vocaset_f16x3.err
export ENC_REPS=10
timeout 600 bash tools/profile_encoders.sh r5_pmc_hubert_bf16_B4 hubert bf16 4 10 > gpurun_out/enc1.log 2>&1
timeout 600 bash tools/profile_encoders.sh r5_pmc_hubert_bf16_B1 hubert bf16 1 10 > gpurun_out/enc2.log 2>&1
timeout 600 bash tools/profile_encoders.sh r5_pmc_hubert_f16x3_B4 hubert f16x3 4 10 > gpurun_out/enc3.log 2>&1
timeout 600 bash tools/profile_encoders.sh r5_pmc_wav2vec_bf16_B1 wav2vec bf16 1 10 > gpurun_out/enc6.log 2>&1
cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/stats_default
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/stats_default -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/stats_default.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/stats_default -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r5_default_bench_cfg2_kernel_stats.csv \;
find $GRAFT_REPO_ROOT/gpurun_out/stats_default -name "*kernel_trace.csv" -delete
cd "$GRAFT_REPO_ROOT"
if [ -z "$SKIP_TESTS" ]; then timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r5_gputests.txt 2>&1; tail -3 gpurun_out/r5_gputests.txt; fi
for i in 1 2 3 6; do grep -m1 "ms per call" gpurun_out/enc$i.log; done
