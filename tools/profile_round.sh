#!/bin/bash
# The round's committed evidence in one GPU job (round 6): counter passes of the benched configuration (tools/pmc_collect.sh: the traced
# run's tuned tiles pinned for every counter pass and recorded), the driver's default bench line, every configuration in every mode
# (tools/bench_all.sh), the reference callers' per-clip workloads (bench.py --config shipped_*), HuBERT-large under rocprofv3
# (tools/profile_encoders.sh), one plain `rocprofv3 --kernel-trace --stats` of the default bench command, and the GPU suite with durations.
# Results land in gpurun_out/ (scratch); the caller copies what is to be judged into profiles/r6_*.
R=${ROUND_TAG:-r6}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/${R}_bench; mkdir -p $O
unset FDM_TILE_OVERRIDE FDM_TUNE
bash tools/pmc_collect.sh ${R}_cfg2_bf16 --dtype bf16 --headline-only 2>&1 | tail -2
unset FDM_TILE_OVERRIDE FDM_TUNE
bash tools/pmc_collect.sh ${R}_cfg2_f16x3 --dtype f16x3 --headline-only 2>&1 | tail -2
unset FDM_TILE_OVERRIDE FDM_TUNE
cd "$GRAFT_REPO_ROOT"
# the committed counter summaries must exist BEFORE the bench lines that quote them are taken
for t in cfg2_bf16 cfg2_f16x3; do mkdir -p profiles/${R}_pmc_$t; cp gpurun_out/pmc_${R}_$t/summary.json gpurun_out/pmc_${R}_$t/summary.md profiles/${R}_pmc_$t/ 2>/dev/null; mkdir -p gpurun_out/profiles_${R}_pmc_$t; cp gpurun_out/pmc_${R}_$t/summary.* gpurun_out/profiles_${R}_pmc_$t/; done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json; echo
bash tools/bench_all.sh 2>&1 | tail -25
for c in shipped_vocaset shipped_mead shipped_biwi cfg1x8; do
  for dt in bf16 f16x3; do
    timeout 900 python bench.py --config $c --dtype $dt --headline-only --steps 3 --warmup 1 --no-cpu-baseline > $O/${c}_$dt.json 2> $O/${c}_$dt.err
    python3 -c "import json; d=json.load(open('$O/${c}_$dt.json')); print('$c', d['dtype'], d['value'], d['ms_per_step'], d.get('stages_ms'), d.get('speedup_vs_sequential_loop'))" || tail -3 $O/${c}_$dt.err
  done
done
export ENC_REPS=10
timeout 600 bash tools/profile_encoders.sh ${R}_pmc_hubert_bf16_B4 hubert bf16 4 10 > gpurun_out/enc1.log 2>&1
grep -m1 "ms per call" gpurun_out/enc1.log
cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/stats_default
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/stats_default -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --headline-only > $GRAFT_REPO_ROOT/gpurun_out/stats_default.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/stats_default -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/${R}_default_bench_cfg2_kernel_stats.csv \;
find $GRAFT_REPO_ROOT/gpurun_out/stats_default -name "*kernel_trace.csv" -delete
cd "$GRAFT_REPO_ROOT"
if [ -z "$SKIP_TESTS" ]; then timeout 1800 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/${R}_gputests.txt 2>&1; tail -20 gpurun_out/${R}_gputests.txt; fi
