#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
bash tools/pmc_collect.sh cfg4_f16x3 --config cfg4 --dtype f16x3 --headline-only > gpurun_out/pmc_cfg4.log 2>&1; tail -2 gpurun_out/pmc_cfg4.log
cat gpurun_out/pmc_cfg4_f16x3/summary.md | cut -c1-200
