#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/pp_probe; mkdir -p $O tools/_build
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DFDM_PP_PHASES $PP_DEFS -o tools/_build/pp_phases tools/pp_probe.cpp 2>/dev/null
for shape in "8192 1024 2048" "6400 1024 2048" "6400 3072 1024" "4096 4096 4096"; do ./tools/_build/pp_phases $shape; done 2>&1 | tee $O/phases${PP_TAG}.txt
