// GEMM kernels for f32 operands (see gemm.hpp).
#include "gemm.hpp"
#include "kernels.hpp"
namespace fdm { hipError_t gemm_launch_f32(const fdm_gemm_args& a, hipStream_t s) { return gemm_dispatch<float>(a, s); } }
