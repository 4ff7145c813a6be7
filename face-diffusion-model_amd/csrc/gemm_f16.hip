// GEMM kernels for single-plane fp16 operands (see gemm.hpp; include/fdm_hip.h FDM_F16).
#include "gemm.hpp"
#include "kernels.hpp"
namespace fdm { hipError_t gemm_launch_f16(const fdm_gemm_args& a, hipStream_t s) { return gemm_dispatch<f16>(a, s); } }
