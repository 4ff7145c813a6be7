#!/bin/bash
# rocprofv3 passes over one once-per-clip stage (tools/bench_encoders.py): kernel trace + stats, then counter passes in
# separate processes (SQ, FETCH_SIZE, WRITE_SIZE).  usage: profile_encoders.sh <tag> <stage> <mode> <B> <seconds>
# output: gpurun_out/<tag>/{stats.csv, sq.csv, fetch.csv, write.csv, wall.txt}; tools/encoder_report.py turns them into summary.md
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
REPS=${ENC_REPS:-10}
python3 $ROOT/tools/bench_encoders.py "$@" 20 > $OUT/wall.txt 2>&1
rm -rf $OUT/raw; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o t -- python3 $ROOT/tools/bench_encoders.py "$@" $REPS > $OUT/trace.log 2>&1
cp $(find $OUT/raw -name "*kernel_stats.csv" | head -1) $OUT/stats.csv; cp $(find $OUT/raw -name "*kernel_trace.csv" | head -1) $OUT/trace.csv; rm -rf $OUT/raw
pass() { local name=$1; shift
  rm -rf $OUT/raw; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/raw -o c -- python3 $ROOT/tools/bench_encoders.py $STAGE_ARGS $REPS > $OUT/$name.log 2>&1
  local f=$(find $OUT/raw -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp $f $OUT/$name.raw.csv; else echo "pass $name: no counters"; tail -3 $OUT/$name.log; fi; rm -rf $OUT/raw; }
STAGE_ARGS="$*"
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE TCC_HIT_sum
pass write WRITE_SIZE TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum
python3 $ROOT/tools/encoder_report.py $OUT "$TAG: bench_encoders.py $*" > $OUT/summary.md
rm -f $OUT/*.raw.csv $OUT/trace.csv
cat $OUT/wall.txt; head -40 $OUT/summary.md
