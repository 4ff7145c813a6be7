// Fused attention kernels for split fp16 operands (see attention.hpp).
#include "attention.hpp"
#include "kernels.hpp"
namespace fdm {
hipError_t attn_launch_f16x3(const fdm_attn_args& a, hipStream_t s) { return attn_launch_dtype<f16x3_t>(a, s); }
}
