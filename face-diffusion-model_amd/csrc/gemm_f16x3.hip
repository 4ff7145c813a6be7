// GEMM kernels for split f16x3 operands (see gemm.hpp, common.hpp Opnd<>).
#include "gemm.hpp"
#include "kernels.hpp"
namespace fdm { hipError_t gemm_launch_f16x3(const fdm_gemm_args& a, hipStream_t s) { return gemm_dispatch_split<f16x3_t>(a, s); } }
