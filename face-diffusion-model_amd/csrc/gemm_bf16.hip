// GEMM kernels for bf16 operands (see gemm.hpp).
#include "gemm.hpp"
#include "kernels.hpp"
namespace fdm { hipError_t gemm_launch_bf16(const fdm_gemm_args& a, hipStream_t s) { return gemm_dispatch<bf16>(a, s); } }
// the FDM_TILE_* a launch with tile = 0 resolves to, for every operand kind (include/fdm_hip.h: fdm_gemm_heuristic_tile)
namespace fdm {
int gemm_heuristic_tile_of(const fdm_gemm_args& a) {
  if (a.ksplit > 1) return gemm_ksplit_heuristic_tile(a);
  if (a.sched_fuse) {        // two forms: 64x64, and the ping-pong tile (one-plane kinds, whole column tiles)
    const bool pp = a.dtype != FDM_F16X3 && gemm_tile_override() == 0 && !a.ln_stat_in && a.N % 128 == 0 && gemm_sched_fuse_heuristic_pp(a, a.dtype == FDM_F32 ? 4 : 2);
    return pp ? FDM_TILE_256x128_PP : FDM_TILE_64x64;
  }
  const int want = gemm_tile_override();
  if (want > 0) return want;
  return gemm_heuristic_tile(a, a.dtype == FDM_F32 ? 4 : 2, a.dtype == FDM_F16X3);
}
}
