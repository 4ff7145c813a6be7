#!/bin/bash
# builds tools/_build/lw_probe_<tile>_<variant> for the tiles the library runs at 1992-3984 rows (see tools/lw_probe.cpp)
cd "$(dirname "$0")/.."; mkdir -p tools/_build
FL="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=13"
build() { # name BM BN WM WN NST LW
  for v in 0 1 2 3 4 5 6 7; do
    /opt/rocm/bin/hipcc $FL -DFDM_LW_VARIANT=$v -DLWP_BM=$2 -DLWP_BN=$3 -DLWP_WM=$4 -DLWP_WN=$5 -DLWP_NST=$6 -DLWP_LW=$7 -o tools/_build/lw_probe_$1_$v tools/lw_probe.cpp 2>&1 | grep -v "argument unused" &
  done; wait
}
build 128x64 128 64 4 2 4 4
build 128x64s3 128 64 4 2 3 4
build 128x128 128 128 2 4 3 4
build 64x64 64 64 2 4 4 4
ls tools/_build | grep -c lw_probe_
# ring depth with loader waves (Little's law: 3 x 16 KB in flight per CU at ~0.5 us of L2 latency is ~96 GB/s): deeper rings, full kernel only
buildv0() { /opt/rocm/bin/hipcc $FL -DFDM_LW_VARIANT=0 -DLWP_BM=$2 -DLWP_BN=$3 -DLWP_WM=$4 -DLWP_WN=$5 -DLWP_NST=$6 -DLWP_LW=$7 -o tools/_build/lw_probe_$1_0 tools/lw_probe.cpp 2>&1 | grep -v "argument unused"; }
buildv0 64x64n6 64 64 2 4 6 4 &
buildv0 64x64n8 64 64 2 4 8 4 &
buildv0 64x64n8l8 64 64 2 4 8 8 &
buildv0 128x64n6 128 64 4 2 6 4 &
buildv0 64x128n4 64 128 2 4 4 4 &
buildv0 64x128n6 64 128 2 4 6 4 &
buildv0 80x128n4 80 128 1 8 4 4 &
buildv0 80x128n5 80 128 1 8 5 4 &
wait
# fewer compute waves with larger wave tiles (fewer fragment bytes read from LDS per staged byte): 2 x 2 compute waves + 4 loaders
buildv0 128x64w22n4 128 64 2 2 4 4 &
buildv0 128x64w22n3 128 64 2 2 3 4 &
buildv0 128x128w22 128 128 2 2 3 4 &
buildv0 64x64w22 64 64 2 2 4 4 &
buildv0 64x128w22 64 128 2 2 4 4 &
buildv0 64x64w22n6 64 64 2 2 6 4 &
buildv0 64x64w22l2 64 64 2 2 4 2 &
wait
