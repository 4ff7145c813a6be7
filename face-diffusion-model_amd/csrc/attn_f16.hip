// Fused attention kernels for single-plane fp16 operands (see attention.hpp; include/fdm_hip.h FDM_F16).
#include "attention.hpp"
#include "kernels.hpp"
namespace fdm { hipError_t attn_launch_f16(const fdm_attn_args& a, hipStream_t s) { return attn_launch_dtype<f16>(a, s); } }
