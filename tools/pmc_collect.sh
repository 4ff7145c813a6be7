#!/bin/bash
# Counter passes over one bench configuration (rocprofv3 --pmc, one pass per process: SQ 8 slots, TCC 4, and never together
# with the trace domains gpurun refuses).  usage: pmc_collect.sh <tag> <bench args...>;  output: gpurun_out/pmc_<tag>/*.csv
# The program goes directly after `--` (python3 bench.py ...; no wrappers: the profiler's library initialises the GPU first).
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
STEPS=${PMC_STEPS:-30}
cd /tmp; export TMPDIR=/tmp
live() { grep -o '"denoiser_steps_per_sample": [0-9]*' $1 | grep -o '[0-9]*$'; }   # DDIM skips its dead last pair: T - 1 live steps
pass() { # name, counters...
  local name=$1; shift
  rm -rf $OUT/raw_$name
  timeout 900 rocprofv3 --pmc "$@" --output-format csv -d $OUT/raw_$name -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --profile-steps $STEPS $BENCH_ARGS > $OUT/$name.log 2>&1
  local f=$(find $OUT/raw_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $ROOT/tools/pmc_report.py reduce $f $OUT/$name.csv $(live $OUT/$name.log) $(grep -o '"kernel_launches_per_diffusion_step": [0-9]*' $OUT/$name.log | grep -o '[0-9]*$'); else echo "pass $name produced no counters"; tail -5 $OUT/$name.log; fi
  rm -rf $OUT/raw_$name
}
BENCH_ARGS="$*"
# durations first (kernel trace only)
rm -rf $OUT/raw_trace
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/raw_trace -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --profile-steps $STEPS $BENCH_ARGS > $OUT/trace.log 2>&1
f=$(find $OUT/raw_trace -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/pmc_report.py trace $f $OUT/trace.csv $(live $OUT/trace.log) $(grep -o '"kernel_launches_per_diffusion_step": [0-9]*' $OUT/trace.log | grep -o '[0-9]*$')
rm -rf $OUT/raw_trace
# The tiles the traced run's tuner picked, per call site ("qkv=11,out=1,..."; bench.py prints them as "gemm_tiles"): the counter
# passes below are separate processes -- pin them to the same set (FDM_TILE_OVERRIDE, no tuning launches) and record the set in the
# summary, so that a bench line can show that its own tiles are the profiled ones (`counters_tiles_match`).
if [ -z "$FDM_TILE_OVERRIDE" ]; then
  TILES=$(python3 -c "
import json, sys
line = [l for l in open('$OUT/trace.log') if l.startswith('{') and 'gemm_tiles' in l][-1]
print(','.join(f'{k}={v}' for k, v in sorted(json.loads(line)['gemm_tiles'].items()) if v))" 2>/dev/null)
  if [ -n "$TILES" ]; then export FDM_TILE_OVERRIDE="$TILES" FDM_TUNE=0; fi
fi
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES
pass fetch FETCH_SIZE TCC_HIT_sum
pass write WRITE_SIZE TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum
pass tcp TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
HEAD=$(cd $ROOT && git rev-parse --short HEAD 2>/dev/null || echo unknown)
python3 $ROOT/tools/pmc_report.py derive $OUT $OUT/summary.json "{\"tag\": \"$TAG\", \"bench_args\": \"$BENCH_ARGS\", \"tiles\": \"${FDM_TILE_OVERRIDE:-heuristic}\", \"steps_profiled\": $(live $OUT/trace.log), \"commit\": \"${FDM_COMMIT:-$HEAD}\", \"date\": \"$(date -u +%F)\"}"
ls $OUT
