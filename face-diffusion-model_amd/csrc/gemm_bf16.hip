// GEMM kernels for bf16 operands (see gemm.hpp).
#include "gemm.hpp"
#include "kernels.hpp"
namespace fdm {
hipError_t gemm_launch_bf16(const fdm_gemm_args& a, hipStream_t s) { return a.lnx_gamma ? gemm_dispatch_lnx<bf16>(a, s) : gemm_dispatch<bf16>(a, s); }
int gemm_lnx_capacity_bf16(int tile, int* bm, int* bn) {
  int cap = 0;
  (void)gemm_lnx_tile<bf16>(tile, 1, nullptr, nullptr, &cap, bm, bn);
  return cap;
}
}  // namespace fdm
