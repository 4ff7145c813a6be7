#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/pp_probe; mkdir -p $O tools/_build
for v in 0 4 6; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DFDM_PP_STAMPS -DFDM_PP_VARIANT=$v $PP_DEFS -o tools/_build/pp_stamps_$v tools/pp_probe.cpp 2>/dev/null &
done
wait
for v in 0 4 6; do echo "== variant $v"; ./tools/_build/pp_stamps_$v 8192 1024 2048; done 2>&1 | tee $O/stamps${PP_TAG}.txt
