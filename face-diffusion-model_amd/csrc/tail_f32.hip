// Fused layer-tail launch for f32 operands (see tail.hpp).
#include "tail.hpp"
#include "kernels.hpp"
namespace fdm { hipError_t tail_launch_f32(const fdm_tail_args& a, hipStream_t s) { return tail_launch_t<float>(a, s); } }
