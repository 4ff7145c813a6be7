// GEMM kernels for split f16x3 operands (see gemm.hpp, common.hpp Opnd<>).
#include "gemm.hpp"
#include "kernels.hpp"
namespace fdm {
hipError_t gemm_launch_f16x3(const fdm_gemm_args& a, hipStream_t s) { return a.lnx_gamma ? gemm_dispatch_lnx<f16x3_t>(a, s) : gemm_dispatch_split<f16x3_t>(a, s); }
int gemm_lnx_capacity_f16x3(int tile, int* bm, int* bn) {
  int cap = 0;
  (void)gemm_lnx_tile<f16x3_t>(tile, 1, nullptr, nullptr, &cap, bm, bn);
  return cap;
}
}  // namespace fdm
