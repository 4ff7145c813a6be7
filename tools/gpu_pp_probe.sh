#!/bin/bash
# builds tools/pp_probe.cpp in its loop-ablation variants on the GPU box and times them
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/pp_probe; mkdir -p $O tools/_build
for v in ${PP_VARIANTS:-0 1 2 3 4 5 6 7}; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DFDM_PP_VARIANT=$v $PP_DEFS -o tools/_build/pp_probe_$v tools/pp_probe.cpp 2>/dev/null &
done
wait
for shape in "8192 1024 2048" "6400 1024 2048"; do
  for v in ${PP_VARIANTS:-0 1 2 3 4 5 6 7}; do ./tools/_build/pp_probe_$v $shape; done
done 2>&1 | tee $O/probe${PP_TAG}.txt
