// GEMM kernels for f32 operands (see gemm.hpp).
#include "gemm.hpp"
#include "kernels.hpp"
namespace fdm {
hipError_t gemm_launch_f32(const fdm_gemm_args& a, hipStream_t s) { return a.lnx_gamma ? gemm_dispatch_lnx<float>(a, s) : gemm_dispatch<float>(a, s); }
int gemm_lnx_capacity_f32(int tile, int* bm, int* bn) {
  int cap = 0;
  (void)gemm_lnx_tile<float>(tile, 1, nullptr, nullptr, &cap, bm, bn);
  return cap;
}
}  // namespace fdm
