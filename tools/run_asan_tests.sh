#!/bin/bash
# CPU suite of the C ABI's host side under AddressSanitizer: argument validation, program recording, plan bookkeeping and
# the host tables (tests that need no device).  GPU ASan is not available on this pool.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -j8 -C "$ROOT/face-diffusion-model_amd/csrc" asan
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd "$ROOT"
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 FDM_LIB_PATH=$ROOT/face-diffusion-model_amd/fdm_amd/libfdm_hip_asan.so \
  python -m pytest tests/test_abi_cpu.py tests/test_plan_host_cpu.py -q -p no:cacheprovider "$@"
