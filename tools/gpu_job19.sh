#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out/r2_job19; mkdir -p $O
cd /tmp
for v in ${VARIANTS:-lnx nowait}; do
  rm -rf $O/raw_$v
  unset FDM_FUSE_LNX FDM_LNX_NOWAIT
  if [ $v = nolnx ]; then export FDM_FUSE_LNX=0; fi
  if [ $v = nowait ]; then export FDM_LNX_NOWAIT=256; fi
  if [ $v = noexch ]; then export FDM_LNX_NOWAIT=512; fi
  if [ $v = ldsonly ]; then export FDM_LNX_NOWAIT=1024; fi
  FDM_TUNE=0 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/raw_$v -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --profile-steps 30 --dtype ${DT:-bf16} > $O/trace_$v.log 2>&1
  f=$(find $O/raw_$v -name "*kernel_trace.csv" | head -1)
  python3 $ROOT/tools/pmc_report.py trace $f $O/trace_$v.csv 30 $(grep -o '"kernel_launches_per_diffusion_step": [0-9]*' $O/trace_$v.log | grep -o '[0-9]*$')
  rm -rf $O/raw_$v
  cat $O/trace_$v.csv | cut -c1-220
done
