#!/bin/bash
# round-2 job 1: launch-floor probe under runtime knobs, f32-mode bench + kernel trace
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job1; mkdir -p $O
for envs in "" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "AMD_DIRECT_DISPATCH=0" "DEBUG_HIP_KERNARG_COPY_OPT=0"; do
  echo "=== env: $envs" >> $O/launch_floor.txt
  env $envs timeout 120 tools/_build/launch_floor >> $O/launch_floor.txt 2>&1
done
timeout 600 python bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err
timeout 600 python bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err
HIP_FORCE_DEV_KERNARG=1 timeout 600 python bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_bf16_devkernarg.json 2> $O/bench_bf16_devkernarg.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_f32 -- python3 $GRAFT_REPO_ROOT/bench.py --dtype f32 --steps 1 --warmup 0 --no-cpu-baseline --profile-steps 30 > $GRAFT_REPO_ROOT/$O/prof_f32.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_f32 -name "*kernel_stats.csv" -exec cp {} $O/prof_f32_kernel_stats.csv \;
find $O/prof_f32 -name "*kernel_trace.csv" -delete
ls -la $O
