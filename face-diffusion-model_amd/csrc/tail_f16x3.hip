// Fused layer-tail launch for f16x3 operands (see tail.hpp).
#include "tail.hpp"
#include "kernels.hpp"
namespace fdm { hipError_t tail_launch_f16x3(const fdm_tail_args& a, hipStream_t s) { return tail_launch_t<f16x3_t>(a, s); } }
