// GEMM kernels for bf16 operands (see gemm.hpp).
#include "gemm.hpp"
#include "kernels.hpp"
namespace fdm { hipError_t gemm_launch_bf16(const fdm_gemm_args& a, hipStream_t s) { return gemm_dispatch<bf16>(a, s); } }
