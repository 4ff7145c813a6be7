// Fused layer-tail launch for bf16 operands (see tail.hpp).
#include "tail.hpp"
#include "kernels.hpp"
namespace fdm { hipError_t tail_launch_bf16(const fdm_tail_args& a, hipStream_t s) { return tail_launch_t<bf16>(a, s); } }
