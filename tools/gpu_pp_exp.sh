#!/bin/bash
# A/B of compile-time experiments on the ping-pong loop: each argument is a set of -D flags
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/pp_probe; mkdir -p $O tools/_build
i=0
for defs in "$@"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off $defs -o tools/_build/pp_exp_$i tools/pp_probe.cpp 2>tools/_build/pp_exp_$i.err &
  i=$((i+1))
done
wait
for round in 1 2; do
  i=0
  for defs in "$@"; do
    echo "[$defs]"; ./tools/_build/pp_exp_$i ${PP_SHAPE:-8192 1024 2048} || tail -3 tools/_build/pp_exp_$i.err
    i=$((i+1))
  done
done 2>&1 | tee -a $O/exp.txt
